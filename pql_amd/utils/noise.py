"""Gaussian action noise with the reference's names (pql/utils/noise.py:19-41)."""
import torch


def _draw(shape, dtype, device, std, generator=None):
    # torch.normal(zeros, std_tensor) consumes the generator as normal_(0,1) then scales (Appendix B of SURVEY.md)
    return torch.empty(shape, dtype=dtype, device=device).normal_(generator=generator) * std


_STD_CACHE = {}


def _fused(x, draw, std_rows, std_scalar, out_bounds, generator):
    """out = clamp(x + draw * sigma, lo, hi) in one launch (plus the draw itself) when x is a contiguous fp32 matrix on the GPU."""
    from pql_amd import _lib as L
    if draw is None:
        draw = torch.empty_like(x).normal_(generator=generator)   # torch.normal(zeros, std) = normal_(0,1) then the scale (Appendix B)
    else:
        draw = draw.to(x.dtype).contiguous()
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        L.check(L.lib.pqlk_action_noise(L.ptr(x), L.ptr(draw), L.ptr(std_rows), float(std_scalar), x.shape[0], x.shape[1],
                                        float(out_bounds[0]), float(out_bounds[1]), L.ptr(out), L.stream(x.device)))
    return out


def _fusable(x, noise_bounds, out_bounds):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous() and x.numel() > 0 and noise_bounds is None
            and out_bounds is not None)


def add_normal_noise(x, std, noise_bounds=None, out_bounds=None, generator=None, draw=None):
    """`draw`: optional injected N(0,1) sample of x's shape (parity tests); otherwise drawn from `generator`."""
    if _fusable(x, noise_bounds, out_bounds) and not torch.is_tensor(std):
        return _fused(x, draw, None, std, out_bounds, generator)
    noise = _draw(x.shape, x.dtype, x.device, std, generator) if draw is None else draw.to(x.dtype) * std
    if noise_bounds is not None:
        noise = noise.clamp(noise_bounds[0], noise_bounds[1])
    out = x + noise
    return out if out_bounds is None else out.clamp(out_bounds[0], out_bounds[1])


def add_mixed_normal_noise(x, std_max, std_min, noise_bounds=None, out_bounds=None, env_offset=0, total_envs=None,
                           generator=None, draw=None):
    """Per-env sigma = linspace(std_min, std_max, N)[env].  env_offset/total_envs let a data-parallel
    rank index the GLOBAL env axis."""
    n = x.shape[0] if total_envs is None else total_envs
    key = (float(std_min), float(std_max), int(n), int(env_offset), int(x.shape[0]), str(x.device))
    std = _STD_CACHE.get(key)
    if std is None:   # built on the host like the reference (same fp32 values), uploaded ONCE: no per-step H2D sync
        std = torch.linspace(std_min, std_max, n)[env_offset: env_offset + x.shape[0]].to(x.device).unsqueeze(-1)
        _STD_CACHE[key] = std
    if _fusable(x, noise_bounds, out_bounds):
        return _fused(x, draw, std, 0.0, out_bounds, generator)
    noise = (_draw(x.shape, x.dtype, x.device, 1.0, generator) if draw is None else draw.to(x.dtype)) * std
    if noise_bounds is not None:
        noise = noise.clamp(noise_bounds[0], noise_bounds[1])
    out = x + noise
    return out if out_bounds is None else out.clamp(out_bounds[0], out_bounds[1])
