"""`identity + dropout(x)` as one HIP pass per direction (csrc/dropout.hip): the tail of every attention / FFN block of
the reference's encoder layers (mmcv: ``self.dropout(output) + identity``).  The keep / drop decision is a hash of
(seed, element index): no mask tensor; the seed comes from torch's CPU generator, so ``torch.manual_seed`` makes a run
reproducible (the masks are not torch's Philox masks — they are not reproducible across implementations in the reference
either)."""
import torch

from ._lib import lib, check, ptr, current_stream


class _DropoutAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, identity, p):
        x = x.contiguous()
        idt = identity.contiguous()
        seed = int(torch.empty((), dtype=torch.int64).random_().item()) & 0x7FFFFFFFFFFFFFFF       # CPU generator: no device sync
        y = torch.empty_like(x)
        check(lib().selfocc_dropout_add_fwd(ptr(x), ptr(idt), ptr(y), x.numel(), float(p), seed, current_stream(x.device)),
              "selfocc_dropout_add_fwd")
        ctx.seed, ctx.p = seed, float(p)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        g = g.contiguous()
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(g)
            check(lib().selfocc_dropout_bwd(ptr(g), ptr(gx), g.numel(), ctx.p, ctx.seed, current_stream(g.device)),
                  "selfocc_dropout_bwd")
        return gx, (g if ctx.needs_input_grad[1] else None), None


def dropout_add(x, identity, p, training):
    """``identity + F.dropout(x, p, training)``.  The fused route needs CUDA float32 tensors of one shape; everything else
    (evaluation, p = 0, CPU, autocast dtypes) is the torch expression."""
    if (training and 0.0 < p < 1.0 and x.is_cuda and x.dtype == torch.float32 and identity.dtype == torch.float32
            and x.shape == identity.shape and x.numel() > 0 and not torch.is_autocast_enabled()):
        return _DropoutAdd.apply(x, identity, p)
    return torch.nn.functional.dropout(x, p, training) + identity
