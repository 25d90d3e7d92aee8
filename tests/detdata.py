"""Deterministic, platform-independent test data.

Every array is a pure integer-hash function of (seed, flat index), evaluated in
uint64/float64 and cast to float32 once, so the golden-vector generator
(tools/gen_golden.py, run in the build container next to the reference) and the
tests (run anywhere, including the GPU box where the reference does not exist)
reconstruct bit-identical inputs and weights without storing them.
"""
import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def _mix(idx, seed):
    """32-bit avalanche hash of idx (uint64 array) keyed by seed; returns uint64 in [0, 2^32)."""
    x = (idx * np.uint64(0x9E3779B1) + np.uint64(seed) * np.uint64(0x85EBCA77) + np.uint64(0x165667B1)) & _M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x2C1B3C6D)) & _M32
    x ^= x >> np.uint64(12)
    x = (x * np.uint64(0x297A2D39)) & _M32
    x ^= x >> np.uint64(15)
    return x


def uniform(shape, seed, lo=-1.0, hi=1.0):
    """float32 array, uniform in [lo, hi), exact function of (shape, seed)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = _mix(np.arange(n, dtype=np.uint64), seed).astype(np.float64) / 4294967296.0
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def bernoulli(shape, seed, p):
    """float32 0/1 array with P(1)=p."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = _mix(np.arange(n, dtype=np.uint64), seed).astype(np.float64) / 4294967296.0
    return (u < p).astype(np.float32).reshape(shape)


def integers(shape, seed, high):
    """int64 array in [0, high)."""
    n = int(np.prod(shape)) if len(shape) else 1
    return (_mix(np.arange(n, dtype=np.uint64), seed) % np.uint64(high)).astype(np.int64).reshape(shape)


def mlp_layer_dims(in_dim, out_dim, hidden=(512, 256, 128)):
    dims = [in_dim, *hidden, out_dim]
    return list(zip(dims[:-1], dims[1:]))


def mlp_state(in_dim, out_dim, seed, hidden=(512, 256, 128), prefix="net."):
    """Reference-keyed state dict (net.{0,2,4,..}.{weight,bias}) of deterministic weights.

    Scale follows the nn.Linear default bound 1/sqrt(fan_in) so activations stay O(1).
    """
    out = {}
    for li, (fi, fo) in enumerate(mlp_layer_dims(in_dim, out_dim, hidden)):
        bound = 1.0 / np.sqrt(fi)
        out[f"{prefix}{2 * li}.weight"] = uniform((fo, fi), seed * 1000 + 2 * li, -bound, bound)
        out[f"{prefix}{2 * li}.bias"] = uniform((fo,), seed * 1000 + 2 * li + 1, -bound, bound)
    return out


def doubleq_state(obs_dim, act_dim, out_dim, seed, hidden=(512, 256, 128)):
    s = {}
    s.update(mlp_state(obs_dim + act_dim, out_dim, seed, hidden, prefix="net_q1.net."))
    s.update(mlp_state(obs_dim + act_dim, out_dim, seed + 1, hidden, prefix="net_q2.net."))
    return s


def bn_critic_state(O, A, seed, hidden=(512, 256, 128)):
    """Reference-keyed state of DoubleQBatchNorm: Linear weights as in doubleq_state (keys 0,3,6,9), non-trivial gamma / beta."""
    st = {}
    for n, pre in enumerate(("net_q1.net.", "net_q2.net.")):
        lin = mlp_state(O + A, 1, seed + n, hidden, prefix="")
        for l in range(len(hidden) + 1):
            st[f"{pre}{3 * l}.weight"] = lin[f"{2 * l}.weight"]; st[f"{pre}{3 * l}.bias"] = lin[f"{2 * l}.bias"]
            if l < len(hidden):
                st[f"{pre}{3 * l + 1}.weight"] = uniform((hidden[l],), 9000 + 10 * (seed + n) + l, 0.5, 1.5)
                st[f"{pre}{3 * l + 1}.bias"] = uniform((hidden[l],), 9500 + 10 * (seed + n) + l, -0.2, 0.2)
    return st


def probe_indices(n, k=256, seed=7):
    """Deterministic sample positions used to summarise a large tensor in a fixture."""
    if n <= k:
        return np.arange(n, dtype=np.int64)
    return np.sort(integers((k,), seed + n, n))


def summarize(arr):
    """Compact fingerprint of a (possibly large) float array: [sum, l2, probes...] in float64."""
    a = np.asarray(arr, dtype=np.float64).reshape(-1)
    p = probe_indices(a.size)
    return np.concatenate([[a.sum(), np.sqrt((a * a).sum())], a[p]])
