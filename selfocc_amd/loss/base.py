import torch.nn as nn

from ..registry import OPENOCC_LOSS


class BaseLoss(nn.Module):
    """weight * loss_func(**{arg: inputs[key]}) with ``input_dict`` = {arg: key} (loss/base_loss.py:9-39)."""

    def __init__(self, weight=1.0, input_dict={'input': 'input'}, **kwargs):
        super().__init__()
        self.weight = weight
        self.input_dict = input_dict
        self.loss_func = lambda: 0
        self.writer = None

    supports_ray_shard = False      # True: the loss reduces LocalRows / tagged per-sample inputs locally (dist.py)

    def forward(self, inputs):
        args = {arg: inputs[key] for arg, key in self.input_dict.items()}
        # ray-sharded head: outputs['ray_shard'] names the shard explicitly (dist.RayShard.local_keys = the per-sample
        # outputs that hold this rank's rows only); LocalRows / tagged tensors carry it as well
        shard = inputs.get('ray_shard') if hasattr(inputs, 'get') else None
        self._ray_shard = shard
        if not self.supports_ray_shard:
            from ..dist import shard_of
            for arg, v in args.items():
                if shard_of(v) is not None or (shard is not None and self.input_dict[arg] in shard.local_keys):
                    raise NotImplementedError(
                        f"{type(self).__name__} received '{arg}' from a ray-sharded head (per-sample tensors of this rank's "
                        f"rows only) but does not reduce them locally; run it with NeuSHead(ray_shard=False)")
        return self.weight * self.loss_func(**args)


@OPENOCC_LOSS.register_module()
class MultiLoss(nn.Module):
    """Sum of the configured losses -> (total, {class name: value}) (loss/multi_loss.py:10-43).

    The reference pays one ``.item()`` host sync per loss per iteration (multi_loss.py:33-41) although train.py reads
    the dict only on print iterations (``f'{loss_value:.5f}'``, train.py:255-263, every ``print_freq``).  Here the
    values stay on the device by default — 0-dim tensors, which format, compare and convert like floats
    (``format(t, '.5f')`` reads the value back at that moment, i.e. only when something is printed): SURVEY §8 f-4.
    ``sync_items=True`` restores python floats."""

    def __init__(self, loss_cfgs, sync_items=False):
        super().__init__()
        assert isinstance(loss_cfgs, list)
        self.num_losses = len(loss_cfgs)
        self.losses = nn.ModuleList([OPENOCC_LOSS.build(cfg) for cfg in loss_cfgs])
        self.iter_counter = 0
        self.sync_items = sync_items

    def forward(self, inputs):
        loss_dict, tot_loss = {}, 0.
        for loss_func in self.losses:
            loss = loss_func(inputs)
            tot_loss = tot_loss + loss
            loss_dict[loss_func.__class__.__name__] = loss.detach().item() if self.sync_items else loss.detach()
        self.iter_counter += 1
        return tot_loss, loss_dict
