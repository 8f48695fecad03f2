from .neus_head import NeuSHead, RaySampler, Img2LiDAR, SDFField
