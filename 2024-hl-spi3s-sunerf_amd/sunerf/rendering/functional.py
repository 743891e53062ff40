"""Glue between the module API and the HIP kernels: forward orchestration and the autograd boundary."""
import torch

from sunerf_hip import ops
from sunerf_hip.train import bucket_of


class _MlpOnPoints(torch.autograd.Function):
    """``NeRF.forward`` on free-standing query points (model.py:44-57) as an autograd node: the fused render kernel fed with the
    points themselves (32 per chunk, no ray, its integral unused), differentiable w.r.t. the model's parameters."""

    @staticmethod
    def forward(ctx, model, points, *params):
        training = any(ctx.needs_input_grad[2:])
        packed = model.packed()
        out = ops.mlp_points_fwd(packed, points, training=training)
        if training:
            ctx.packed, ctx.params, ctx.n_padded = packed, params, out['n_padded']
            ctx.param_meta = [(p.shape, p.device) for p in params]
            ctx.save_for_backward(out['stash'], points)
        return out['raw'][:, :packed.d_out] if packed.d_out < 2 else out['raw']

    @staticmethod
    def backward(ctx, g_raw):
        stash, points = ctx.saved_tensors
        if points.shape[0] != ctx.n_padded:      # the kernels work on whole 32-point chunks: zero points with zero gradient
            points = torch.cat([points, points.new_zeros(ctx.n_padded - points.shape[0], 4)])
        query = ('points', points)
        g = g_raw.new_zeros(ctx.n_padded, 2)
        g[:g_raw.shape[0], :g_raw.shape[1]] = g_raw
        g = g.view(ctx.n_padded // 32, 32, 2)
        absmax = g.abs().max().reshape(1).view(torch.int32)      # bit pattern of max |g_raw| (sunerf_common.h: gradient scale)
        direct = _grad_targets(ctx.params)
        if direct is not None:
            ops.mlp_backward(ctx.packed, g, absmax, stash, direct[0], direct[1], accumulate=True, query=query)
            _announce(ctx.params)
            return (None,) * (2 + len(ctx.params))
        gW = [torch.empty(shape, dtype=torch.float32, device=dev) for shape, dev in ctx.param_meta[0::2]]
        gb = [torch.empty(shape, dtype=torch.float32, device=dev) for shape, dev in ctx.param_meta[1::2]]
        ops.mlp_backward(ctx.packed, g, absmax, stash, gW, gb, query=query)
        grads = []
        for w, b in zip(gW, gb):
            grads += [w, b]
        return (None,) * 2 + tuple(grads)


def mlp_points(model, x: torch.Tensor) -> torch.Tensor:
    """NeRF.forward on arbitrary query points (M, 4) -> (M, d_out) (model.py:44-57): the fused kernel's free-standing-points
    mode (``sunerf_mlp_points_fwd``), every lane of it a query point.  Differentiable w.r.t. the model's parameters like the
    reference's module call (a loss on free-standing points trains)."""
    flat = x.reshape(-1, 4)
    params = []
    for lin in model.linears():
        params += [lin.weight, lin.bias]
    if torch.is_grad_enabled() and any(p.requires_grad for p in params):
        return _MlpOnPoints.apply(model, flat, *params)
    packed = model.packed()
    raw = ops.mlp_points_fwd(packed, flat)['raw']
    return raw[:, :packed.d_out] if packed.d_out < 2 else raw


def _grad_targets(params):
    """(grad_weights, grad_biases) views to accumulate into directly, or None.

    When every parameter already owns a contiguous fp32 ``.grad`` on its device (``ClipAdam`` / ``GradBucket`` keep them as
    views of one flat buffer), the weight-gradient kernel adds its result straight into it and the autograd node reports "no
    gradient" for the parameters -- instead of returning 18 fresh tensors per model that autograd would then add to
    ``.grad`` with 18 tiny kernels (a third of the step at the reference's default batch of 1024 rays)."""
    grads = []
    for p in params:
        g = p.grad
        if g is None or g.dtype != torch.float32 or g.device != p.device or not g.is_contiguous() or g.shape != p.shape:
            return None
        grads.append(g)
    return grads[0::2], grads[1::2]


def _scalar_head_slice(scalars):
    """The slice of a flat gradient bucket that the given 0-d parameters occupy back to back (``ClipAdam`` tags every parameter with
    ``(owner, offset, numel)``), or None: lets eight scalar gradients be accumulated with one launch instead of eight."""
    tags = [bucket_of(p) for p in scalars]
    if any(t is None for t in tags):
        return None
    owner, first = tags[0][0], tags[0][1]
    for i, (p, (o, off, k)) in enumerate(zip(scalars, tags)):
        if o is not owner or off != first + i or k != 1:
            return None
        # the parameter's .grad must still BE its slot of the bucket: a replaced / cleared .grad is copied over (or zeroed
        # into) the slot by the optimiser's step, which would lose what is added here
        if not p.requires_grad or p.grad is None or p.grad.data_ptr() != owner.flat_grads[off:off + 1].data_ptr():
            return None
    return owner.flat_grads[first:first + len(tags)]


def _announce(params):
    """The gradients of ``params`` are final in their flat bucket: let its owner start the all-reduce of that slice while the
    other model's backward still runs (``ClipAdam(overlap=True)``)."""
    owner = bucket_of(params[0])
    if owner is not None:
        owner[0].segment_ready(params)


class _EmissionPass(torch.autograd.Function):
    """One fused render pass (coarse or fine) as an autograd node.

    Differentiable outputs: ``image`` and ``regularization`` (the two the training loss of sunerf.py:110-120 uses);
    gradients are produced for the MLP parameters only -- the reference's graph has no path to the rays
    (sampling.py:120 detaches the resampled z).  ``weights`` / ``absorption`` / maps are marked non-differentiable."""

    @staticmethod
    def forward(ctx, model, rays_o, rays_d, times, z_vals, reg_radius, want_epilogues, *params):
        training = any(ctx.needs_input_grad[7:])
        ctx.set_materialize_grads(False)      # unused / non-differentiable outputs: None instead of (N,S) zero tensors
        packed = model.packed()
        out = ops.emission_render_fwd(packed, rays_o, rays_d, times, z_vals, reg_radius,
                                      want_epilogues=want_epilogues, training=training)
        ctx.training = training
        if training:
            ctx.packed = packed
            ctx.reg_radius = reg_radius
            ctx.n_params = len(params)
            ctx.params = params
            ctx.param_meta = [(p.shape, p.device) for p in params]
            ctx.save_for_backward(rays_o, rays_d, z_vals, out['raw'], out['stash'], times)
        outs = [out['image'], out['weights'], out['absorption']]
        non_diff = [out['weights'], out['absorption']]
        if want_epilogues:
            outs += [out['height_map'], out['absorption_map'], out['regularization']]
            non_diff += [out['height_map'], out['absorption_map']]
        ctx.mark_non_differentiable(*non_diff)
        ctx.want_epilogues = want_epilogues
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_image, g_weights, g_absorption, g_hm=None, g_am=None, g_reg=None):
        rays_o, rays_d, z_vals, raw, stash, times = ctx.saved_tensors
        n, s = z_vals.shape
        if g_image is None and g_reg is None:
            return (None,) * (7 + ctx.n_params)
        if g_image is None:
            g_image = torch.zeros(n, dtype=torch.float32, device=z_vals.device)
        # parameters arrive as (W0, b0, W1, b1, ...)
        direct = _grad_targets(ctx.params)
        if direct is not None:
            ops.emission_render_bwd(ctx.packed, rays_o, rays_d, z_vals, raw, stash, g_image, g_reg, 0.0, ctx.reg_radius,
                                    direct[0], direct[1], accumulate=True, times=times)
            _announce(ctx.params)
            return (None,) * (7 + ctx.n_params)
        gW = [torch.empty(shape, dtype=torch.float32, device=dev) for shape, dev in ctx.param_meta[0::2]]
        gb = [torch.empty(shape, dtype=torch.float32, device=dev) for shape, dev in ctx.param_meta[1::2]]
        ops.emission_render_bwd(ctx.packed, rays_o, rays_d, z_vals, raw, stash, g_image, g_reg, 0.0, ctx.reg_radius,
                                gW, gb, times=times)
        grads = []
        for w, b in zip(gW, gb):
            grads += [w, b]
        return (None,) * 7 + tuple(grads)


def emission_pass(model, rays_o, rays_d, times, z_vals, reg_radius, want_epilogues):
    """Dict of one pass' outputs; goes through autograd when gradients are enabled and the model is trainable."""
    params = []
    for lin in model.linears():
        params += [lin.weight, lin.bias]
    if torch.is_grad_enabled() and any(p.requires_grad for p in params):
        outs = _EmissionPass.apply(model, rays_o, rays_d, times, z_vals, reg_radius, want_epilogues, *params)
        keys = ['image', 'weights', 'absorption'] + (['height_map', 'absorption_map', 'regularization']
                                                     if want_epilogues else [])
        return dict(zip(keys, outs))
    return ops.emission_render_fwd(model.packed(), rays_o, rays_d, times, z_vals, reg_radius,
                                   want_epilogues=want_epilogues)


class _MlpOnRays(torch.autograd.Function):
    """``NeRF.forward`` (model.py:44-57) on the samples ``o + d z`` of a ray batch as an autograd node: the raw network output
    (N, S, d_output), differentiable w.r.t. the model's parameters.  This is what the generic ``SuNeRFRendering._render``
    (base_tracing.py:118-129) hands to a subclass's ``raw2outputs``: the MLP runs in the fused render kernel (whose own
    integral outputs are ignored) and its backward in the data / weight gradient kernels, fed with whatever gradient the
    subclass's torch code sends back."""

    @staticmethod
    def forward(ctx, model, rays_o, rays_d, times, z_vals, *params):
        training = any(ctx.needs_input_grad[5:])
        packed = model.packed()
        out = ops.emission_render_fwd(packed, rays_o, rays_d, times, z_vals, 0.0, want_raw=True, training=training)
        ctx.training = training
        if training:
            ctx.packed, ctx.params = packed, params
            ctx.param_meta = [(p.shape, p.device) for p in params]
            ctx.save_for_backward(out['stash'], rays_o, rays_d, times, z_vals)
        ctx.d_out = packed.d_out
        return out['raw'][..., :packed.d_out] if packed.d_out < 2 else out['raw']

    @staticmethod
    def backward(ctx, g_raw):
        stash, rays_o, rays_d, times, z_vals = ctx.saved_tensors
        query = ('rays', rays_o, rays_d, times, z_vals)
        if g_raw.shape[-1] < 2:
            g_raw = torch.cat([g_raw, torch.zeros_like(g_raw)], -1)
        g_raw = g_raw.contiguous().float()
        # bit pattern of max |g_raw|: the scale the fp16 backward arithmetic is normalised with (sunerf_common.h)
        absmax = g_raw.abs().max().reshape(1).view(torch.int32)
        direct = _grad_targets(ctx.params)
        if direct is not None:
            ops.mlp_backward(ctx.packed, g_raw, absmax, stash, direct[0], direct[1], accumulate=True, query=query)
            _announce(ctx.params)
            return (None,) * (5 + len(ctx.params))
        gW = [torch.empty(shape, dtype=torch.float32, device=dev) for shape, dev in ctx.param_meta[0::2]]
        gb = [torch.empty(shape, dtype=torch.float32, device=dev) for shape, dev in ctx.param_meta[1::2]]
        ops.mlp_backward(ctx.packed, g_raw, absmax, stash, gW, gb, query=query)
        grads = []
        for w, b in zip(gW, gb):
            grads += [w, b]
        return (None,) * 5 + tuple(grads)


def mlp_on_rays(model, rays_o, rays_d, times, z_vals) -> torch.Tensor:
    """(N, S, d_output) raw output of ``model`` (a ``NeRF``) at the samples of the rays; goes through autograd when the
    model is trainable and gradients are enabled."""
    params = []
    for lin in model.linears():
        params += [lin.weight, lin.bias]
    if torch.is_grad_enabled() and any(p.requires_grad for p in params):
        return _MlpOnRays.apply(model, rays_o, rays_d, times, z_vals, *params)
    raw = ops.emission_render_fwd(model.packed(), rays_o, rays_d, times, z_vals, 0.0, want_raw=True)['raw']
    return raw[..., :model.packed().d_out] if model.packed().d_out < 2 else raw


class _EmissionIntegral(torch.autograd.Function):
    """``EmissionRadiativeTransfer.raw2outputs`` (emission.py:14-54) on a given raw tensor: differentiable w.r.t. ``raw`` through
    all three outputs (image, weights, regularizing_quantity), like the reference's autograd graph."""

    @staticmethod
    def forward(ctx, raw, z_vals, rays_d):
        image, weights, absorption = ops.emission_integral_fwd(raw.detach(), z_vals, rays_d)
        ctx.save_for_backward(raw.detach(), z_vals, rays_d)
        ctx.set_materialize_grads(False)
        return image, weights, absorption

    @staticmethod
    def backward(ctx, g_image, g_weights, g_absorption):
        raw, z_vals, rays_d = ctx.saved_tensors
        if g_image is None and g_weights is None and g_absorption is None:
            return None, None, None
        return ops.emission_integral_bwd(raw, z_vals, rays_d, g_image, g_weights, g_absorption), None, None


def emission_raw2outputs(raw, z_vals, rays_d):
    image, weights, absorption = _EmissionIntegral.apply(raw, z_vals, rays_d)
    return {'image': image, 'weights': weights, 'regularizing_quantity': absorption}


class _DtIntegral(torch.autograd.Function):
    """``DensityTemperatureRadiativeTransfer.raw2outputs`` (density_temperature.py:192-271) on given inferences (base offsets
    already added, as ``NeRF_DT.forward`` returns them).  Differentiable through ``image`` w.r.t. the inferences, the seven
    absorption scalars and the volumetric constant (what the loss of sunerf.py:187-195 needs)."""

    @staticmethod
    def forward(ctx, tables, pixel_factor, inferences, z_vals, rays_d, wavelengths, vol_c, *la):
        la_vec = torch.stack([p.detach() for p in la])
        zeros = torch.zeros_like(rays_d)
        out = ops.dt_integral_fwd(inferences.detach(), z_vals, zeros, rays_d, wavelengths, tables[0], tables[1], la_vec, vol_c, 0.0, 0.0,
                                  pixel_factor, 0.0)
        ctx.tables, ctx.pixel_factor = tables, pixel_factor
        ctx.save_for_backward(inferences.detach(), z_vals, rays_d, wavelengths, la_vec, vol_c.detach())
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(out['weights'], out['reg_q'])
        return out['image'], out['weights'], out['reg_q']

    @staticmethod
    def backward(ctx, g_image, g_weights, g_q):
        inferences, z_vals, rays_d, wavelengths, la_vec, vol_c = ctx.saved_tensors
        if g_image is None:
            return (None,) * (8 + la_vec.shape[0])
        g_raw, g_la, g_vc, _ = ops.dt_integral_bwd(inferences, z_vals, torch.zeros_like(rays_d), rays_d, wavelengths, ctx.tables[0],
                                                   ctx.tables[1], la_vec, vol_c, 0.0, 0.0, ctx.pixel_factor, 0.0,
                                                   g_image.contiguous(), None)
        return (None, None, g_raw, None, None, None, g_vc.reshape(())) + tuple(g_la[i] for i in range(g_la.shape[0]))


def dt_raw2outputs(tables, pixel_factor, inferences, log_abs, vol_c, z_vals, rays_d, wavelengths):
    la = [log_abs[str(w)] for w in ops.AIA_WAVELENGTHS]
    image, weights, reg_q = _DtIntegral.apply(tables, pixel_factor, inferences, z_vals, rays_d, wavelengths, vol_c, *la)
    return {'image': image, 'weights': weights, 'regularizing_quantity': reg_q}


class _DtPass(torch.autograd.Function):
    """One fused density/temperature pass: render kernel (MLP) -> DT integral kernel.  Differentiable outputs: ``image``
    (N,W) and ``regularization``; gradients for the MLP parameters, the 7 ``log_absortpion`` scalars and
    ``volumetric_constant``."""

    @staticmethod
    def forward(ctx, model, tables, pixel_factor, rays_o, rays_d, times, z_vals, wavelengths, reg_radius, want_epilogues,
                vol_c, *params):
        n_la = len(ops.AIA_WAVELENGTHS)
        la = torch.stack([p.detach() for p in params[:n_la]])
        training = any(ctx.needs_input_grad[10:])
        ctx.set_materialize_grads(False)
        packed = model.packed()
        # (the DT image goes with rho^2 = exp(2 raw_0): twice the emission image's sensitivity to the raw output)
        mlp = ops.emission_render_fwd(packed, rays_o, rays_d, times, z_vals, 0.0, want_raw=True, training=training,
                                      probe_sensitivity=2.0)
        out = ops.dt_integral_fwd(mlp['raw'], z_vals, rays_o, rays_d, wavelengths, tables[0], tables[1], la, vol_c,
                                  model.base_log_density, model.base_log_temperature, pixel_factor, reg_radius,
                                  want_epilogues=want_epilogues)
        if training:
            ctx.packed, ctx.tables, ctx.pixel_factor, ctx.reg_radius = packed, tables, pixel_factor, reg_radius
            ctx.base = (model.base_log_density, model.base_log_temperature)
            ctx.param_meta = [(p.shape, p.device) for p in params[n_la:]]
            ctx.mlp_params = params[n_la:]
            ctx.scalar_params = tuple(params[:n_la]) + (vol_c,)
            ctx.save_for_backward(rays_o, rays_d, z_vals, wavelengths, mlp['raw'], mlp['stash'], la, vol_c.detach(), times)
        outs = [out['image'], out['weights'], out['reg_q']]
        non_diff = [out['weights'], out['reg_q']]
        if want_epilogues:
            outs += [out['height_map'], out['absorption_map'], out['regularization']]
            non_diff += [out['height_map'], out['absorption_map']]
        ctx.mark_non_differentiable(*non_diff)
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_image, g_weights, g_q, g_hm=None, g_am=None, g_reg=None):
        rays_o, rays_d, z_vals, wavelengths, raw, stash, la, vol_c, times = ctx.saved_tensors
        query = ('rays', rays_o, rays_d, times, z_vals)
        if g_image is None:
            g_image = torch.zeros(z_vals.shape[0], wavelengths.shape[1], dtype=torch.float32, device=z_vals.device)
        g_raw, g_la, g_vc, absmax = ops.dt_integral_bwd(raw, z_vals, rays_o, rays_d, wavelengths, ctx.tables[0], ctx.tables[1],
                                                        la, vol_c, ctx.base[0], ctx.base[1], ctx.pixel_factor, ctx.reg_radius,
                                                        g_image.contiguous(), g_reg)
        la_slice = _scalar_head_slice(ctx.scalar_params[:-1])
        vol_c_param = ctx.scalar_params[-1]
        vc_tag = bucket_of(vol_c_param)
        if (la_slice is not None and vc_tag is not None and vol_c_param.requires_grad and vol_c_param.grad is not None
                and vol_c_param.grad.data_ptr() == vc_tag[0].flat_grads[vc_tag[1]:vc_tag[1] + 1].data_ptr()):
            # the seven absorption scalars sit back to back in the optimiser's flat gradient buffer (the volumetric constant, a
            # direct parameter of the module, elsewhere in it): two adds instead of eight AccumulateGrad launches
            la_slice.add_(g_la)
            vol_c_param.grad.add_(g_vc.reshape(vol_c_param.grad.shape))
            head = (None,) * (11 + g_la.shape[0])
        else:
            head = (None,) * 10 + (g_vc.reshape(()),) + tuple(g_la[i] for i in range(g_la.shape[0]))
        direct = _grad_targets(ctx.mlp_params)
        if direct is not None:
            ops.mlp_backward(ctx.packed, g_raw, absmax, stash, direct[0], direct[1], accumulate=True, query=query)
            _announce(ctx.mlp_params)
            return head + (None,) * len(ctx.mlp_params)
        gW = [torch.empty(shape, dtype=torch.float32, device=dev) for shape, dev in ctx.param_meta[0::2]]
        gb = [torch.empty(shape, dtype=torch.float32, device=dev) for shape, dev in ctx.param_meta[1::2]]
        ops.mlp_backward(ctx.packed, g_raw, absmax, stash, gW, gb, query=query)
        grads = []
        for w, b in zip(gW, gb):
            grads += [w, b]
        return head + tuple(grads)


def dt_pass(model, tables, pixel_factor, rays_o, rays_d, times, z_vals, wavelengths, reg_radius, want_epilogues):
    """Dict of one DT pass' outputs (image (N,W), weights, regularizing_quantity[, maps, regularization])."""
    if hasattr(model, 'field_on_rays'):
        # analytic field (SimpleStar) instead of an MLP: same integral, inference only (stellar_model.py, image_render.py:266)
        with torch.no_grad():
            raw = model.field_on_rays(rays_o, rays_d, z_vals)
            la = torch.stack([model.log_absortpion[str(w)].detach() for w in ops.AIA_WAVELENGTHS])
            out = ops.dt_integral_fwd(raw, z_vals, rays_o, rays_d, wavelengths, tables[0], tables[1], la,
                                      model.volumetric_constant, model.base_log_density, model.base_log_temperature,
                                      pixel_factor, reg_radius, want_epilogues=want_epilogues)
        out['regularizing_quantity'] = out.pop('reg_q')
        return out
    la = [model.log_absortpion[str(w)] for w in ops.AIA_WAVELENGTHS]
    mlp_params = []
    for lin in model.linears():
        mlp_params += [lin.weight, lin.bias]
    outs = _DtPass.apply(model, tables, pixel_factor, rays_o, rays_d, times, z_vals, wavelengths, reg_radius, want_epilogues,
                         model.volumetric_constant, *la, *mlp_params)
    keys = ['image', 'weights', 'regularizing_quantity'] + (['height_map', 'absorption_map', 'regularization']
                                                            if want_epilogues else [])
    return dict(zip(keys, outs))
