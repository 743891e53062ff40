"""TEST INFRASTRUCTURE ONLY -- CPU restatement (PyTorch fp32, CPU tensors) of the reference's hot path.

This is the *oracle* the HIP path is checked against.  It is never imported by the product package
(``2024-hl-spi3s-sunerf_amd/``); only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` use it.  It is pinned against the real reference by ``tests/test_oracle_golden.py`` using
the fixtures under ``tests/golden/`` that ``oracle/gen_golden.py`` produced by running the reference's own
code in the build container (the reference's tree has no tests or golden vectors of its own,
SURVEY.md section 4).

Every function cites the reference lines (relative to /root/reference) it restates.  The aten op order
is kept identical to the reference so the restatement is bit-exact against it on CPU; the structure
(pure functions over explicit parameter lists instead of nn.Modules) is our own.

Defects D1/D2 of the reference at HEAD are resolved to their evident intent:
the MLP output dict is unwrapped and the regularisation is (N, S).
"""
import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch

Params = List[Tuple[torch.Tensor, torch.Tensor]]  # [(W[out,in], b[out]) ...]: in, hidden..., out


# --------------------------------------------------------------------------------------------------------------
# samplers -- sunerf/train/sampling.py
# --------------------------------------------------------------------------------------------------------------
def linspace_t_vals(n_samples: int) -> torch.Tensor:
    """sampling.py:65-66 -- the (1, S) buffer; must be torch.linspace's own fp32 values."""
    return torch.linspace(0., 1., n_samples)[None].to(torch.float32)


def _jitter(z_vals: torch.Tensor, t_rand: torch.Tensor) -> torch.Tensor:
    """sampling.py:93-98 (and :44-49): uniform sample inside the bin around each z."""
    mids = .5 * (z_vals[:, 1:] + z_vals[:, :-1])
    upper = torch.concat([mids, z_vals[:, -1:]], dim=1)
    lower = torch.concat([z_vals[:, :1], mids], dim=1)
    return lower + (upper - lower) * t_rand


def stratified_z(rays_o, rays_d, t_vals, distance: torch.Tensor, solar_R: torch.Tensor,
                 t_rand: Optional[torch.Tensor] = None) -> torch.Tensor:
    """StratifiedSampler.forward, sampling.py:68-91 (+ :93-98 when ``t_rand`` is given).

    ``distance`` and ``solar_R`` are the fp32 0-d buffers of sampling.py:62-63
    (``distance / Rs_per_ds`` and ``1 / Rs_per_ds``)."""
    dist_o = rays_o.pow(2).sum(-1).pow(0.5)
    a = rays_d.pow(2).sum(-1)
    b = (2 * rays_o * rays_d).sum(-1)
    c = rays_o.pow(2).sum(-1) - solar_R ** 2
    dist_inner = (-b - torch.sqrt(b.pow(2) - 4 * a * c)) / (2 * a)
    dist_near = dist_o - distance
    dist_far = dist_o + distance
    hit = ~torch.isnan(dist_inner)
    dist_far = torch.where(hit, dist_inner, dist_far)
    z_vals = dist_near[:, None] * (1. - t_vals) + dist_far[:, None] * t_vals
    if t_rand is not None:
        z_vals = _jitter(z_vals, t_rand)
    return z_vals


def spherical_z(rays_o, rays_d, t_vals, distance: torch.Tensor, solar_R: torch.Tensor,
                t_rand: Optional[torch.Tensor] = None) -> torch.Tensor:
    """SphericalSampler.forward, sampling.py:16-42 (+ :44-49)."""
    a = rays_d.pow(2).sum(-1)
    b = (2 * rays_o * rays_d).sum(-1)
    c = rays_o.pow(2).sum(-1) - distance ** 2
    dist_near = (-b - torch.sqrt(b.pow(2) - 4 * a * c)) / (2 * a)
    dist_far = (-b + torch.sqrt(b.pow(2) - 4 * a * c)) / (2 * a)
    c = rays_o.pow(2).sum(-1) - solar_R ** 2
    dist_inner = (-b - torch.sqrt(b.pow(2) - 4 * a * c)) / (2 * a)
    hit = ~torch.isnan(dist_inner)
    dist_far = torch.where(hit, dist_inner, dist_far)
    z_vals = dist_near[:, None] * (1. - t_vals) + dist_far[:, None] * t_vals
    if t_rand is not None:
        z_vals = _jitter(z_vals, t_rand)
    return z_vals


def points_on_rays(rays_o, rays_d, z_vals) -> torch.Tensor:
    """sampling.py:100 / :52 / :123 -- product rounded, then the sum (no FMA)."""
    return rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]


def sample_pdf(bins: torch.Tensor, weights: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
    """HierarchicalSampler.sample_pdf, sampling.py:128-169: inverse-transform samples of the piecewise-constant density
    ``weights`` (N, B-1) on ``bins`` (N, B) at the CDF positions ``u`` ((S,) shared or (N, S) per ray)."""
    pdf = (weights + 1e-5) / torch.sum(weights + 1e-5, -1, keepdim=True)
    cdf = torch.cumsum(pdf, dim=-1)
    cdf = torch.concat([torch.zeros_like(cdf[..., :1]), cdf], dim=-1)
    if u.dim() == 1:
        u = u.expand(list(cdf.shape[:-1]) + [u.shape[0]])
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    cdf_lo = torch.gather(cdf, -1, below)
    cdf_hi = torch.gather(cdf, -1, above)
    bin_lo = torch.gather(bins, -1, below)
    bin_hi = torch.gather(bins, -1, above)
    denom = cdf_hi - cdf_lo
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_lo) / denom
    return bin_lo + t * (bin_hi - bin_lo)


def hierarchical_z(z_vals: torch.Tensor, weights: torch.Tensor, n_samples: int,
                   u: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """HierarchicalSampler.forward, sampling.py:111-126, around :func:`sample_pdf`.

    Returns ``(new_z_samples (N, n_samples), z_vals_combined (N, S + n_samples))``.  ``u`` replaces the
    deterministic ``linspace`` of :139-141 when the caller wants the ``perturb=True`` branch (:143)."""
    bins = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
    if u is None:
        u = torch.linspace(0., 1., n_samples, device=z_vals.device)
    new_z = sample_pdf(bins, weights[..., 1:-1], u).detach()
    z_comb, _ = torch.sort(torch.cat([z_vals, new_z], dim=-1), dim=-1)
    return new_z, z_comb


# --------------------------------------------------------------------------------------------------------------
# field model -- sunerf/model/model.py
# --------------------------------------------------------------------------------------------------------------
def positional_encoding(x: torch.Tensor, n_freqs: int = 10, scale_factor: float = 2.) -> torch.Tensor:
    """PositionalEncoding.forward, model.py:123-132 (freq bands :113-114): (M, 4) -> (M, 84).

    Layout: 4 raw, 40 sin (frequency-major, coordinate-minor), 40 cos."""
    freq_bands = 2. ** torch.linspace(0., n_freqs - 1, n_freqs)
    f = freq_bands[None, :, None]
    arg = x[:, None, :] * f / scale_factor
    return torch.concat([x, torch.sin(arg).reshape(x.shape[0], -1), torch.cos(arg).reshape(x.shape[0], -1)],
                        dim=-1)


def mlp_forward(params: Params, x: torch.Tensor, encoding: bool = True,
                return_hidden: bool = False):
    """NeRF.forward, model.py:44-57 (Sine with w0 = 1, :66-72): encode, 8x(Linear+sin), Linear."""
    h = positional_encoding(x) if encoding else x
    hidden = []
    for W, b in params[:-1]:
        h = torch.sin(1. * torch.nn.functional.linear(h, W, b))
        if return_hidden:
            hidden.append(h)
    W, b = params[-1]
    out = torch.nn.functional.linear(h, W, b)
    return (out, hidden) if return_hidden else out


def mlp_forward_half(params: Params, x: torch.Tensor) -> torch.Tensor:
    """The MLP as the HALF arithmetic evaluates it (include/sunerf_hip.h, SUNERF_PRECISION_HALF): every operand of every
    product rounded to fp16 -- the encoded input, the sine outputs, and the weights AFTER the 1/(2 pi) that the pack kernel
    folds into the non-output layers (fp32 multiply, then round) -- products and sums exact (float64 here; fp32 on the
    matrix cores), biases fp32, sin evaluated on the pre-activation in revolutions.  This is the emulated-low-precision
    oracle of SURVEY.md section 8d ("bf16 configs: vs emulated oracle at 1e-4, deviation from fp32 reported")."""
    inv2pi = torch.tensor(0.15915494309189535, dtype=torch.float32)
    h = positional_encoding(x).to(torch.float16).to(torch.float64)
    for W, b in params[:-1]:
        w16 = (W.to(torch.float32) * inv2pi).to(torch.float16).to(torch.float64)
        rev = h @ w16.T + (b.to(torch.float32) * inv2pi).to(torch.float64)
        h = torch.sin(2 * math.pi * rev).to(torch.float32).to(torch.float16).to(torch.float64)
    W, b = params[-1]
    return (h @ W.to(torch.float16).to(torch.float64).T + b.to(torch.float64)).to(torch.float32)


def init_params(d_filter: int = 256, n_layers: int = 8, d_in: int = 84, d_out: int = 2,
                seed: int = 7) -> Params:
    """Default nn.Linear initialisation in the creation order of NeRF.__init__ (model.py:28-42):
    in_layer, layers[0..n_layers-2], out_layer."""
    g = torch.Generator().manual_seed(seed)
    dims = [d_in] + [d_filter] * n_layers + [d_out]
    params = []
    for fan_in, fan_out in zip(dims[:-1], dims[1:]):
        bound = 1. / math.sqrt(fan_in)
        W = (torch.rand(fan_out, fan_in, generator=g) * 2 - 1) * bound
        b = (torch.rand(fan_out, generator=g) * 2 - 1) * bound
        params.append((W, b))
    return params


def params_from_state_dict(sd: Dict[str, torch.Tensor], prefix: str) -> Params:
    """State-dict key names of the reference module tree:
    ``{prefix}in_layer.1.{weight,bias}``, ``{prefix}layers.{i}.{weight,bias}``, ``{prefix}out_layer.*``."""
    params = [(sd[f'{prefix}in_layer.1.weight'], sd[f'{prefix}in_layer.1.bias'])]
    i = 0
    while f'{prefix}layers.{i}.weight' in sd:
        params.append((sd[f'{prefix}layers.{i}.weight'], sd[f'{prefix}layers.{i}.bias']))
        i += 1
    params.append((sd[f'{prefix}out_layer.weight'], sd[f'{prefix}out_layer.bias']))
    return params


# --------------------------------------------------------------------------------------------------------------
# emission / absorption integral -- sunerf/rendering/emission.py, base_tracing.py
# --------------------------------------------------------------------------------------------------------------
def cumprod_exclusive(t: torch.Tensor) -> torch.Tensor:
    """base_tracing.py:135-156."""
    c = torch.cumprod(t, -1)
    c = torch.roll(c, 1, -1)
    c[..., 0] = 1.
    return c


def emission_integral(raw: torch.Tensor, z_vals: torch.Tensor, rays_d: torch.Tensor) -> Dict[str, torch.Tensor]:
    """EmissionRadiativeTransfer.raw2outputs, emission.py:14-54."""
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists[..., :1], dists], dim=-1)
    dists = dists * torch.norm(rays_d[..., None, :], dim=-1)
    intensity = torch.exp(raw[..., 0]) * dists
    absorption = torch.exp(-torch.nn.functional.relu(raw[..., 1]) * dists)
    total_absorption = cumprod_exclusive(absorption + 1e-10)
    emerging = intensity * total_absorption
    image = emerging.sum(1)[:, None]
    weights = emerging / (emerging.sum(1)[:, None] + 1e-10)
    return {'image': image, 'weights': weights, 'regularizing_quantity': absorption}


def render_pass(params: Params, rays_o, rays_d, times, z_vals, half: bool = False, encoding: bool = True) -> Dict[str, torch.Tensor]:
    """One coarse or fine pass: time concat (base_tracing.py:64-65, :83-84), ``_render`` (:118-129, D1
    resolved) and the emission integral.  ``half``: the MLP in the emulated HALF arithmetic (``mlp_forward_half``)."""
    pts = points_on_rays(rays_o, rays_d, z_vals)
    exp_times = times[:, None].repeat(1, pts.shape[1], 1)
    query = torch.cat([pts, exp_times], -1)
    if half:
        raw = mlp_forward_half(params, query.view(-1, 4))
    else:
        raw = mlp_forward(params, query.view(-1, 4), encoding=encoding)     # encoding=False: NeRF(encoding=None), model.py:32-33
    raw = raw.reshape(*query.shape[:-1], -1)
    out = emission_integral(raw, z_vals, rays_d)
    out['raw'] = raw
    out['points'] = pts
    return out


def render_emission(coarse: Params, fine: Params, rays_o, rays_d, times, *, Rs_per_ds: float = 1.,
                    n_coarse: int = 64, n_fine: int = 128, distance: float = 1.3,
                    sampler: str = 'stratified', t_vals: Optional[torch.Tensor] = None,
                    t_rand: Optional[torch.Tensor] = None,
                    z_vals_combined: Optional[torch.Tensor] = None, encoding: bool = True) -> Dict[str, torch.Tensor]:
    """SuNeRFRendering.forward, base_tracing.py:46-111, for the emission subclass.

    ``z_vals_combined`` lets a test feed the fine pass with externally supplied sample positions
    (stage-wise parity: the inverse-CDF step amplifies tiny weight differences)."""
    t_vals = linspace_t_vals(n_coarse) if t_vals is None else t_vals
    dist_buf = torch.tensor(distance / Rs_per_ds, dtype=torch.float32)
    solar_R = torch.tensor(1 / Rs_per_ds, dtype=torch.float32)
    zfn = stratified_z if sampler == 'stratified' else spherical_z
    z_vals = zfn(rays_o, rays_d, t_vals, dist_buf, solar_R, t_rand)
    c = render_pass(coarse, rays_o, rays_d, times, z_vals, encoding=encoding)
    new_z, z_comb = hierarchical_z(z_vals, c['weights'], n_fine)
    if z_vals_combined is not None:
        z_comb = z_vals_combined
    f = render_pass(fine, rays_o, rays_d, times, z_comb, encoding=encoding)
    absorption = f['regularizing_quantity']
    dist_pts = f['points'].pow(2).sum(-1).pow(0.5)
    return {
        'z_vals_stratified': z_vals,
        'coarse_image': c['image'],
        'z_vals_hierarchical': new_z,
        'fine_image': f['image'],
        'image': f['image'],
        'height_map': (f['weights'] * dist_pts).sum(-1),
        'absorption_map': (1 - absorption).sum(-1),
        # base_tracing.py:43-44 with D2 resolved to (N, S)
        'regularization': torch.relu(dist_pts - 1.2 / Rs_per_ds) * (1 - absorption),
        # extras (not reference outputs) used by stage-wise tests
        '_coarse_weights': c['weights'], '_coarse_raw': c['raw'], '_fine_raw': f['raw'],
        '_z_vals_combined': z_comb, '_fine_weights': f['weights'],
    }


# --------------------------------------------------------------------------------------------------------------
# training-step epilogue -- sunerf/model/sunerf.py, sunerf/train/scaling.py
# --------------------------------------------------------------------------------------------------------------
def asinh_scaling(image: torch.Tensor, vmax: float = 1., a: float = 0.005) -> torch.Tensor:
    """ImageAsinhScaling.forward, scaling.py:17-28 (normalisation constant rounded to fp32 as there)."""
    import numpy as np
    normalization = torch.tensor(np.arcsinh(1 / a), dtype=torch.float32)
    a_t = torch.tensor(a, dtype=torch.float32)
    vmax_t = torch.tensor(vmax, dtype=torch.float32)
    image = image / vmax_t
    return torch.asinh(image / a_t) / normalization


def emission_training_loss(outputs: Dict[str, torch.Tensor], target_image: torch.Tensor,
                           lambda_image: float = 1., lambda_regularization: float = 1.,
                           vmax: float = 1., a: float = 0.005) -> Dict[str, torch.Tensor]:
    """EmissionSuNeRFModule.training_step, sunerf.py:110-120."""
    mse = torch.nn.MSELoss()
    target = asinh_scaling(target_image, vmax, a)
    coarse_loss = mse(asinh_scaling(outputs['coarse_image'], vmax, a), target)
    fine_loss = mse(asinh_scaling(outputs['fine_image'], vmax, a), target)
    reg_loss = outputs['regularization'].mean()
    loss = lambda_image * (coarse_loss + fine_loss) + lambda_regularization * reg_loss
    return {'loss': loss, 'coarse': coarse_loss, 'fine': fine_loss, 'regularization': reg_loss}


def clip_grad_norm(grads: List[torch.Tensor], max_norm: float):
    """torch.nn.utils.clip_grad_norm_ (Lightning's gradient_clip_val, run_emission.py:72): global L2 norm over all
    tensors, coefficient max_norm / (norm + 1e-6) clamped to 1.  Returns (total_norm, clipped grads)."""
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads]))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return total, [g * coef for g in grads]


def adam_step(params: List[torch.Tensor], grads: List[torch.Tensor], exp_avg: List[torch.Tensor],
              exp_avg_sq: List[torch.Tensor], step: int, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8):
    """torch.optim.Adam (sunerf.py:31; no weight decay / amsgrad), operation order of torch/optim/adam.py's
    single-tensor path.  ``step`` counts from 1.  Updates the lists in place."""
    import math
    beta1, beta2 = betas
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    step_size = lr / bc1
    bc2_sqrt = math.sqrt(bc2)
    for i, g in enumerate(grads):
        exp_avg[i] = exp_avg[i] + (g - exp_avg[i]) * (1 - beta1)
        exp_avg_sq[i] = exp_avg_sq[i] * beta2 + (1 - beta2) * g * g
        denom = exp_avg_sq[i].sqrt() / bc2_sqrt + eps
        params[i] = params[i] + (-step_size) * exp_avg[i] / denom


def lr_after(steps: int, start: float = 1e-4, end: float = 1e-5, iterations: float = 1e6, floor: float = 5e-5) -> float:
    """sunerf.py:32-40: ExponentialLR(gamma = (end / start) ** (1 / iterations)) stepped after every batch while the last
    learning rate is above ``floor``."""
    gamma = (end / start) ** (1 / iterations)
    lr = start
    for _ in range(steps):
        if lr > floor:
            lr = lr * gamma
    return lr


# --------------------------------------------------------------------------------------------------------------
# synthetic inputs -- restates data/ray_sampling.py:11-35 and
# train/coordinate_transformation.py:36-54 without sunpy/astropy
# --------------------------------------------------------------------------------------------------------------
def pose_spherical(theta: float, phi: float, radius: float, shift=None) -> torch.Tensor:
    """coordinate_transformation.py:36-54: camera-to-world 4x4 from (lon, lat, radius), including the axis permutation of
    :50 and the optional shift of :51-52."""
    import numpy as np
    T = lambda rows: torch.Tensor(rows).float()   # noqa: E731  (the reference builds its matrices the same way)
    c2w = T([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    c2w = T([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1]]) @ c2w
    c2w = T([[1, 0, 0, 0], [0, np.cos(phi), -np.sin(phi), 0], [0, np.sin(phi), np.cos(phi), 0], [0, 0, 0, 1]]) @ c2w
    c2w = T([[np.cos(theta), 0, -np.sin(theta), 0], [0, 1, 0, 0], [np.sin(theta), 0, np.cos(theta), 0], [0, 0, 0, 1]]) @ c2w
    c2w = T([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]]) @ c2w
    if shift is not None:
        tx, ty, tz = shift
        c2w = T([[1, 0, 0, tx], [0, 1, 0, ty], [0, 0, 1, tz], [0, 0, 0, 1]]) @ c2w
    return c2w


def get_rays(Tx, Ty, c2w) -> Tuple[torch.Tensor, torch.Tensor]:
    """ray_sampling.py:7-36 for helioprojective angles Tx, Ty [rad] (float64 arrays of shape (H, W)): numpy, like the
    reference.  Returns (H, W, 3) origins and directions as fp32 tensors."""
    import numpy as np
    Tx, Ty = np.asarray(Tx, dtype=np.float64), np.asarray(Ty, dtype=np.float64)
    c2w = np.asarray(c2w, dtype=np.float32)
    x = np.sin(Tx)
    y = -np.sin(Ty) * np.cos(Tx)
    z = -np.cos(Tx) * np.cos(Ty)
    directions = np.stack([x, y, z], axis=-1, dtype=np.float32)
    rays_d = np.sum(directions[..., None, :] * c2w[:3, :3], axis=-1)
    rays_o = np.tile(c2w[None, :3, -1], [rays_d.shape[0], rays_d.shape[1], 1])
    return torch.from_numpy(rays_o), torch.from_numpy(rays_d)


def synthetic_rays(resolution: int, theta: float = -0.3, phi: float = 0.1, radius: float = 215.032,
                   fov_half_rad: float = 1.1 * 960. / 206264.806) -> Tuple[torch.Tensor, torch.Tensor]:
    """ray_sampling.py:11-35 on a square helioprojective grid: (R*R, 3) origins and directions, fp32."""
    c2w = pose_spherical(theta, phi, radius)
    lin = torch.linspace(-fov_half_rad, fov_half_rad, resolution, dtype=torch.float64)
    Ty, Tx = torch.meshgrid(lin, lin, indexing='ij')
    x = torch.sin(Tx)
    y = -torch.sin(Ty) * torch.cos(Tx)
    zc = -torch.cos(Tx) * torch.cos(Ty)
    directions = torch.stack([x, y, zc], -1).to(torch.float32)
    rays_d = torch.sum(directions[..., None, :] * c2w[:3, :3], dim=-1).reshape(-1, 3)
    rays_o = c2w[:3, -1].expand(rays_d.shape).contiguous()
    return rays_o, rays_d.contiguous()


# --------------------------------------------------------------------------------------------------------------
# density / temperature head -- sunerf/rendering/density_temperature.py, sunerf/model/model.py:136-187
# --------------------------------------------------------------------------------------------------------------
AIA_WAVELENGTHS = (94, 131, 171, 193, 211, 304, 335)


def read_aia_response_genx(path: str):
    """The AIA temperature-response table the reference loads with ``sunpy.io.special.read_genx``
    (density_temperature.py:131).  IDL ``savegen`` / XDR big-endian file: every channel structure ``A<wl>`` holds
    ``LOGTE float32[101]`` followed by ``TRESP float64[101]`` starting 32 bytes after the second occurrence of its
    4-byte padded tag (offsets 1184, 2436, ... verified against the file length 9908 in SURVEY.md section 8a-6).
    Returns (logte float32 [7, 101], tresp float64 [7, 101])."""
    import numpy as np
    raw = open(path, 'rb').read()
    logte, tresp = [], []
    for wl in AIA_WAVELENGTHS:
        tag = ('A%d' % wl).encode().ljust(4, b'\x00')
        first = raw.find(tag)
        second = raw.find(tag, first + 4)
        off = second + 32
        lt = np.frombuffer(raw, dtype='>f4', count=101, offset=off).astype(np.float32)
        tr = np.frombuffer(raw, dtype='>f8', count=101, offset=off + 404).astype(np.float64)
        assert abs(lt[0] - 4.0) < 1e-6 and abs(lt[-1] - 9.0) < 1e-5 and np.all(np.diff(lt) > 0), 'unexpected genx layout'
        logte.append(lt)
        tresp.append(tr)
    return np.stack(logte), np.stack(tresp)


def interp1d_linear_extrap0(x: torch.Tensor, y: torch.Tensor, xq: torch.Tensor) -> torch.Tensor:
    """Restatement of ``xitorch.interpolate.Interp1D(x, y, method='linear', extrap=0)`` (density_temperature.py:144-146):
    piecewise-linear inside [x[0], x[-1]], 0 outside.  xitorch is not installed and no reference test pins it:
    PARITY UNPINNED for this sub-step (from-documentation semantics)."""
    idx = torch.searchsorted(x, xq.contiguous(), right=True) - 1
    idx = idx.clamp(0, x.numel() - 2)
    x0, x1, y0, y1 = x[idx], x[idx + 1], y[idx], y[idx + 1]
    val = y0 + (xq - x0) * (y1 - y0) / (x1 - x0)
    inside = (xq >= x[0]) & (xq <= x[-1])
    return torch.where(inside, val, torch.zeros_like(val))


def dt_integral(inferences: torch.Tensor, log_abs: Dict[str, torch.Tensor], vol_c: torch.Tensor, z_vals: torch.Tensor,
                wavelengths: torch.Tensor, logte: torch.Tensor, resp: torch.Tensor,
                pixel_intensity_factor: float) -> Dict[str, torch.Tensor]:
    """DensityTemperatureRadiativeTransfer.raw2outputs, density_temperature.py:192-271 (the unused cm-``dists`` of
    :223-232 dropped, defect D4).  ``logte`` (7,101) / ``resp`` (7,101) fp32 = table x aia_exp_time, as :137-146 build."""
    wl = wavelengths[:, None, :].expand(wavelengths.shape[0], inferences.shape[1], wavelengths.shape[1])
    density = torch.exp(torch.nn.functional.relu(inferences[..., 0]))
    density = density[:, :, None].expand(-1, -1, wl.shape[2])
    log_temperature = torch.nn.functional.relu(inferences[..., 1])
    log_temperature = log_temperature[:, :, None].expand(-1, -1, wl.shape[2])
    temperature_response = torch.zeros_like(log_temperature)
    absorption_coefficients = torch.zeros_like(wl).float()
    for c, w in enumerate(AIA_WAVELENGTHS):
        sel = wl == float(w)
        if sel.any():
            tmp = interp1d_linear_extrap0(logte[c], resp[c], log_temperature.flatten()).reshape(temperature_response.shape)
            temperature_response = torch.where(sel, tmp, temperature_response)
            absorption_coefficients = torch.where(sel, torch.nn.functional.relu(log_abs[str(w)]).expand_as(wl), absorption_coefficients)
    absorption = density * absorption_coefficients
    absorption_integral = torch.cumulative_trapezoid(absorption, x=z_vals[:, :, None], dim=1)
    emission = density.pow(2) * temperature_response
    pixel_intensity_term = torch.exp(-absorption_integral) * emission[:, 0:-1, :]
    pixel_intensity = torch.trapezoid(pixel_intensity_term, x=z_vals[:, 0:-1, None], dim=1) * vol_c * pixel_intensity_factor
    weights = torch.nn.functional.relu(inferences[..., 0])
    weights = weights / (weights.sum(1)[:, None] + 1e-10)
    return {'image': pixel_intensity, 'weights': weights,
            'regularizing_quantity': torch.nn.functional.relu(inferences[..., 0])}


def render_pass_dt(params: Params, log_abs, vol_c, rays_o, rays_d, times, z_vals, wavelengths, logte, resp,
                   pixel_intensity_factor, base_log_density: float = 10.0, base_log_temperature: float = 5.0):
    """DT ``_render`` (density_temperature.py:148-190) with NeRF_DT.forward (model.py:169-187)."""
    pts = points_on_rays(rays_o, rays_d, z_vals)
    query = torch.cat([pts, times[:, None].repeat(1, pts.shape[1], 1)], -1)
    x = mlp_forward(params, query.view(-1, 4))
    x = torch.stack([x[:, 0] + base_log_density, x[:, 1] + base_log_temperature], -1)
    inferences = x.reshape(*query.shape[:-1], -1)
    out = dt_integral(inferences, log_abs, vol_c, z_vals, wavelengths, logte, resp, pixel_intensity_factor)
    out['inferences'] = inferences
    out['points'] = pts
    return out


def simple_star_field(query_points: torch.Tensor, rho_0, h0, T0, Rs, t_photosphere: float = 5777.) -> torch.Tensor:
    """SimpleStar.forward, stellar_model.py:53-102: (M, >=3) points -> (M, 2) = (ln rho, log10 T).  ``rho_0, h0, T0, Rs``
    are the fp32 ``stellar_parameters`` (0-dim tensors or floats)."""
    f = lambda v: torch.as_tensor(v, dtype=torch.float32)   # noqa: E731
    rho_0, h0, T0, Rs = f(rho_0), f(h0), f(T0), f(Rs)
    x, y, z = query_points[:, 0], query_points[:, 1], query_points[:, 2]
    radius = torch.sqrt(x ** 2 + y ** 2 + z ** 2)
    rho = torch.zeros_like(radius)
    temp = torch.zeros_like(radius)
    inner, outer = radius <= 1.0, radius > 1.0
    rho[inner] = rho_0
    rho[outer] = rho_0 * torch.exp(1 / h0 * (1 / radius[outer] - 1))
    rho = torch.log(rho)
    temp[inner] = t_photosphere
    ramp = torch.logical_and(radius > 1, radius <= Rs)
    temp[ramp] = (radius[ramp] - 1) * ((T0 - t_photosphere) / (Rs - 1)) + t_photosphere
    temp[radius > Rs] = T0
    return torch.stack((rho, torch.log10(temp)), dim=-1)


def render_dt_analytic(field, log_abs, vol_c, rays_o, rays_d, wavelengths, logte, resp, *, Rs_per_ds=1., n_coarse=64,
                       n_fine=128, distance=1.3, pixel_intensity_factor=1e10, t_vals=None) -> Dict[str, torch.Tensor]:
    """SuNeRFRendering.forward for DensityTemperatureRadiativeTransfer(model=SimpleStar) (image_render.py:266-268):
    ``field`` maps (M, 3) points to (M, 2) inferences and plays both the coarse and the fine model."""
    t_vals = linspace_t_vals(n_coarse) if t_vals is None else t_vals
    z_vals = stratified_z(rays_o, rays_d, t_vals, torch.tensor(distance / Rs_per_ds, dtype=torch.float32),
                          torch.tensor(1 / Rs_per_ds, dtype=torch.float32))

    def one_pass(z):
        pts = points_on_rays(rays_o, rays_d, z)
        inferences = field(pts.reshape(-1, 3)).reshape(*pts.shape[:-1], 2)
        out = dt_integral(inferences, log_abs, vol_c, z, wavelengths, logte, resp, pixel_intensity_factor)
        out['points'] = pts
        return out
    c = one_pass(z_vals)
    new_z, z_comb = hierarchical_z(z_vals, c['weights'], n_fine)
    f = one_pass(z_comb)
    q = f['regularizing_quantity']
    dist_pts = f['points'].pow(2).sum(-1).pow(0.5)
    return {'z_vals_stratified': z_vals, 'coarse_image': c['image'], 'z_vals_hierarchical': new_z, 'fine_image': f['image'],
            'image': f['image'], 'height_map': (f['weights'] * dist_pts).sum(-1), 'absorption_map': (1 - q).sum(-1),
            'regularization': torch.relu(dist_pts - 1.25 / Rs_per_ds) * torch.relu(q)}


def render_dt(coarse: Params, fine: Params, log_abs_c, vol_c_c, log_abs_f, vol_c_f, rays_o, rays_d, times, wavelengths,
              logte, resp, *, Rs_per_ds=1., n_coarse=64, n_fine=128, distance=1.3, pixel_intensity_factor=1e10,
              t_vals=None) -> Dict[str, torch.Tensor]:
    """SuNeRFRendering.forward (base_tracing.py:46-111) for the DT subclass (regularization: density_temperature.py:273-274)."""
    t_vals = linspace_t_vals(n_coarse) if t_vals is None else t_vals
    z_vals = stratified_z(rays_o, rays_d, t_vals, torch.tensor(distance / Rs_per_ds, dtype=torch.float32),
                          torch.tensor(1 / Rs_per_ds, dtype=torch.float32))
    c = render_pass_dt(coarse, log_abs_c, vol_c_c, rays_o, rays_d, times, z_vals, wavelengths, logte, resp, pixel_intensity_factor)
    new_z, z_comb = hierarchical_z(z_vals, c['weights'], n_fine)
    f = render_pass_dt(fine, log_abs_f, vol_c_f, rays_o, rays_d, times, z_comb, wavelengths, logte, resp, pixel_intensity_factor)
    q = f['regularizing_quantity']
    dist_pts = f['points'].pow(2).sum(-1).pow(0.5)
    return {'z_vals_stratified': z_vals, 'coarse_image': c['image'], 'z_vals_hierarchical': new_z, 'fine_image': f['image'],
            'image': f['image'], 'height_map': (f['weights'] * dist_pts).sum(-1), 'absorption_map': (1 - q).sum(-1),
            'regularization': torch.relu(dist_pts - 1.25 / Rs_per_ds) * torch.relu(q),
            '_z_vals_combined': z_comb, '_fine_inferences': f['inferences'], '_coarse_weights': c['weights']}
