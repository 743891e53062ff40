"""Mirror of the reference's ``sunerf/model/sunerf.py``: the Lightning-module API surface on the fused renderer.

``pytorch_lightning`` is optional: when it is importable the modules subclass ``LightningModule`` (so
``run_emission.py`` works unchanged); otherwise a minimal base with the same hook names is used and
``fit_steps`` below drives ``training_step`` / ``configure_optimizers`` / ``on_train_batch_end`` directly.
"""
import os

import torch
from torch import nn
from torch.optim.lr_scheduler import ExponentialLR

from sunerf.rendering.base_tracing import SuNeRFRendering
from sunerf.rendering.emission import EmissionRadiativeTransfer
from sunerf.rendering.density_temperature import DensityTemperatureRadiativeTransfer
from sunerf.train.scaling import ImageAsinhScaling
from sunerf_hip.train import ClipAdam, env_flag, training_loss

try:  # pragma: no cover - depends on the environment
    from pytorch_lightning import LightningModule
except Exception:  # ModuleNotFoundError in this image
    class LightningModule(nn.Module):
        """Stand-in with the hooks the reference modules use (log is a dict; no trainer)."""

        def __init__(self):
            super().__init__()
            self.logged = {}

        def log(self, name, value, **kwargs):
            self.logged[name] = value


class BaseSuNeRFModule(LightningModule):
    """sunerf.py:15-59."""

    def __init__(self, Rs_per_ds, seconds_per_dt, rendering: SuNeRFRendering, validation_dataset_mapping=None,
                 lr_config=None):
        super().__init__()
        self.Rs_per_ds = Rs_per_ds
        self.seconds_per_dt = seconds_per_dt
        self.rendering = rendering
        self.validation_dataset_mapping = validation_dataset_mapping
        self.validation_outputs = {}
        self.lr_config = {'start': 1e-4, 'end': 1e-5, 'iterations': 1e6} if lr_config is None else lr_config

    def configure_optimizers(self):
        # sunerf.py:31: Adam(lr = start).  ClipAdam is the same update on one flat buffer (and can take the gradient
        # clip that run_emission.py:72 configures on the Trainer: fit_steps below sets optimizer.max_norm).
        self.optimizer = ClipAdam(self.rendering.parameters(), lr=self.lr_config['start'])
        self.scheduler = ExponentialLR(self.optimizer, gamma=(self.lr_config['end'] / self.lr_config['start']) ** (
                1 / self.lr_config['iterations']))
        return [self.optimizer], [self.scheduler]

    # sunerf.py:105-107 asserts on NaN / Inf in every step, which costs a device -> host round trip.  True (default)
    # keeps that behaviour with ONE 4-byte read per step; False defers the check to ``check_finite()`` (the optimiser
    # step is skipped on the device for a step that saw non-finite outputs, so the weights stay intact meanwhile).
    strict_finite_check = True

    def _finish_step(self, loss, stats):
        self.last_stats = stats
        # under data parallelism the assert must be taken by every rank together (a rank that raised alone would leave the
        # others waiting in the all-reduce): there it is made after the optimiser step, on the all-reduced count
        if self.strict_finite_check and not _data_parallel():
            self.check_finite()
        self.log('loss', loss)
        self.log('train', {'coarse': stats[1], 'fine': stats[2], 'regularization': stats[3], 'psnr': stats[4]})
        return loss

    def check_finite(self, optimizer=None):
        """The reference's NaN / Inf assert (sunerf.py:105-107) as one 4-byte read: of this rank's counter, or -- given
        the optimiser after its step -- of the count summed over all ranks."""
        count = optimizer.nonfinite if optimizer is not None else getattr(self, 'last_stats', [None] * 6)[5]
        if count is not None:
            assert float(count) == 0, '! [Numerical Alert] an output contains NaN or Inf.'

    def on_train_batch_end(self, *args, **kwargs):
        # Under data parallelism the reference's NaN / Inf assert (sunerf.py:105-107) is taken HERE, after the optimiser step, on
        # the count the gradient all-reduce has summed over the ranks -- by every driver that ends a batch through this hook
        # (Lightning's closure-driven automatic optimisation as well as fit_steps below); _finish_step cannot do it before the
        # step without leaving the other ranks alone in the collective.
        optimizer = getattr(self, 'optimizer', None)
        if self.strict_finite_check and _data_parallel() and optimizer is not None:
            self.check_finite(optimizer)
        if self.strict_finite_check:
            from sunerf_hip import ops as _ops
            _ops.pipe_status()          # a pipelined backward launch that gave up (csrc/bwd_pipe.hip) is reported, not hidden
        if self.scheduler.get_last_lr()[0] > 5e-5:
            self.scheduler.step()
        self.log('Learning Rate', self.scheduler.get_last_lr()[0])

    def validation_epoch_end(self, outputs_list):
        """sunerf.py:42-54: concatenates the per-batch dicts of every validation set and files them under the set's name.
        Accepts a list of batch dicts (one validation set) or a list of such lists; nothing is stored when any set is empty."""
        per_set = outputs_list
        if per_set and isinstance(per_set[0], dict):
            per_set = [per_set]
        if not per_set or not all(len(batches) for batches in per_set):
            if per_set:
                self.validation_outputs = {}
            return
        self.validation_outputs = {
            self.validation_dataset_mapping[i]: {key: torch.cat([b[key] for b in batches]) for key in batches[0]}
            for i, batches in enumerate(per_set)}

    def on_load_checkpoint(self, checkpoint):
        """sunerf.py:56-59: non-strict restore (checkpoints written before a module gained a buffer still load)."""
        self.validation_outputs = {}
        self.load_state_dict(checkpoint['state_dict'], strict=False)


def _data_parallel() -> bool:
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _other_outputs(outputs):
    """Outputs that only take part in the finite check (the images and the regularization are read by the loss anyway)."""
    skip = ('coarse_image', 'fine_image', 'regularization', 'image')     # 'image' is fine_image (base_tracing.py:107)
    return [v for k, v in outputs.items() if k not in skip]


def save_state(sunerf: BaseSuNeRFModule, data_module, save_path):
    """sunerf.py:62-74: pickles the rendering module + data configuration (the ``.snf`` file)."""
    output_path = '/'.join(save_path.split('/')[0:-1])
    os.makedirs(output_path, exist_ok=True)
    torch.save({'rendering': sunerf.rendering, 'data_config': data_module.config, 'Rs_per_ds': data_module.Rs_per_ds,
                'seconds_per_dt': data_module.seconds_per_dt, 'ref_time': data_module.ref_time}, save_path)


class EmissionSuNeRFModule(BaseSuNeRFModule):
    """sunerf.py:77-149."""

    def __init__(self, Rs_per_ds, seconds_per_dt, image_scaling_config, lambda_image=1.0, lambda_regularization=1.0,
                 sampling_config=None, hierarchical_sampling_config=None, model_config=None, **kwargs):
        self.lambda_image = lambda_image
        self.lambda_regularization = lambda_regularization
        rendering = EmissionRadiativeTransfer(Rs_per_ds=Rs_per_ds, sampling_config=sampling_config,
                                              hierarchical_sampling_config=hierarchical_sampling_config,
                                              model_config=model_config)
        super().__init__(Rs_per_ds=Rs_per_ds, seconds_per_dt=seconds_per_dt, rendering=rendering, **kwargs)
        self.image_scaling = ImageAsinhScaling(**image_scaling_config)
        self.mse_loss = nn.MSELoss()

    def _asinh_constants(self):
        """(vmax, a) of the image scaling as host floats, read from the module's buffers once (not per step: a device ->
        host copy each)."""
        key = (self.image_scaling.vmax.data_ptr(), self.image_scaling.vmax._version, self.image_scaling.a._version)
        if getattr(self, '_asinh_key', None) != key:
            self._asinh_key = key
            self._asinh = (float(self.image_scaling.vmax), float(self.image_scaling.a))
        return self._asinh

    def training_step(self, batch, batch_nb):
        tracing = batch['tracing']
        rays, time, target_image = tracing['rays'], tracing['time'], tracing['target_image']
        rays_o, rays_d = rays[:, 0].contiguous(), rays[:, 1].contiguous()
        outputs = self.rendering(rays_o, rays_d, time)
        # sunerf.py:105-125 in one kernel: finite check of all outputs, asinh scaling, 2 x MSE, regularization mean, psnr
        loss, stats = training_loss(outputs['coarse_image'], outputs['fine_image'], target_image.reshape(-1, 1),
                                    outputs['regularization'], self.lambda_image, self.lambda_regularization,
                                    asinh_scaling=self._asinh_constants(),
                                    finite_check=_other_outputs(outputs))
        return self._finish_step(loss, stats)

    def validation_step(self, batch, batch_nb, **kwargs):
        dataloader_idx = kwargs['dataloader_idx'] if 'dataloader_idx' in kwargs else 0
        if dataloader_idx == 0:
            rays, time, target_image = batch['rays'], batch['time'], batch['target_image']
            rays_o, rays_d = rays[:, 0].contiguous(), rays[:, 1].contiguous()
            with torch.no_grad():
                outputs = self.rendering(rays_o, rays_d, time)
            distance = rays_o.pow(2).sum(-1).pow(0.5)
            return {'target_image': target_image, 'fine_image': outputs['fine_image'],
                    'coarse_image': outputs['coarse_image'], 'height_map': outputs['height_map'],
                    'absorption_map': outputs['absorption_map'], 'z_vals_stratified': outputs['z_vals_stratified'],
                    'z_vals_hierarchical': outputs['z_vals_hierarchical'], 'distance': distance}


class DensityTemperatureSuNeRFModule(BaseSuNeRFModule):
    """sunerf.py:152-224."""

    def __init__(self, Rs_per_ds, seconds_per_dt, image_scaling_config, model, loss=nn.MSELoss(), lambda_image=1.0,
                 lambda_regularization=1.0, sampling_config=None, hierarchical_sampling_config=None,
                 pixel_intensity_factor=1e17, model_config=None, **kwargs):
        self.lambda_image = lambda_image
        self.lambda_regularization = lambda_regularization
        rendering_kwargs = {k: kwargs.pop(k) for k in ('response_table', 'response_path') if k in kwargs}
        rendering = DensityTemperatureRadiativeTransfer(Rs_per_ds=Rs_per_ds, sampling_config=sampling_config,
                                                        hierarchical_sampling_config=hierarchical_sampling_config,
                                                        model_config=model_config, model=model,
                                                        pixel_intensity_factor=pixel_intensity_factor, **rendering_kwargs)
        super().__init__(Rs_per_ds=Rs_per_ds, seconds_per_dt=seconds_per_dt, rendering=rendering, **kwargs)
        self.loss = loss

    def training_step(self, batch, batch_nb):
        tracing = batch['tracing']
        rays, time, target_image, wavelengths = (tracing['rays'], tracing['time'], tracing['target_image'],
                                                 tracing['wavelength'])
        rays_o, rays_d = rays[:, 0].contiguous(), rays[:, 1].contiguous()
        outputs = self.rendering.forward(rays_o, rays_d, time, wavelengths)
        if not isinstance(self.loss, nn.MSELoss) or self.loss.reduction != 'mean':
            return self._training_step_generic_loss(outputs, target_image)
        loss, stats = training_loss(outputs['coarse_image'], outputs['fine_image'], target_image,
                                    outputs['regularization'], self.lambda_image, self.lambda_regularization,
                                    asinh_scaling=None, finite_check=_other_outputs(outputs))
        return self._finish_step(loss, stats)

    def _training_step_generic_loss(self, outputs, target_image):
        """A user-supplied loss module other than nn.MSELoss (sunerf.py:159 takes any callable): torch ops."""
        finite = torch.stack([torch.isfinite(v).all() for v in outputs.values()]).all()
        assert bool(finite), '! [Numerical Alert] an output contains NaN or Inf.'
        coarse_loss = self.loss(outputs['coarse_image'], target_image)
        fine_loss = self.loss(outputs['fine_image'], target_image)
        regularization_loss = outputs['regularization'].mean()
        loss = (self.lambda_image * (coarse_loss + fine_loss) + self.lambda_regularization * regularization_loss)
        with torch.no_grad():
            psnr = -10. * torch.log10(fine_loss)
        self.log('loss', loss)
        self.log('train', {'coarse': coarse_loss, 'fine': fine_loss, 'regularization': regularization_loss, 'psnr': psnr})
        return loss

    def validation_step(self, batch, batch_nb, **kwargs):
        dataloader_idx = kwargs['dataloader_idx'] if 'dataloader_idx' in kwargs else 0
        if dataloader_idx == 0:
            rays, time, target_image, wavelengths = batch['rays'], batch['time'], batch['target_image'], batch['wavelength']
            rays_o, rays_d = rays[:, 0].contiguous(), rays[:, 1].contiguous()
            with torch.no_grad():
                outputs = self.rendering(rays_o, rays_d, time, wavelengths)
            distance = rays_o.pow(2).sum(-1).pow(0.5)
            return {'target_image': target_image, 'fine_image': outputs['fine_image'],
                    'coarse_image': outputs['coarse_image'], 'height_map': outputs['height_map'],
                    'absorption_map': outputs['absorption_map'], 'z_vals_stratified': outputs['z_vals_stratified'],
                    'z_vals_hierarchical': outputs['z_vals_hierarchical'], 'distance': distance}


def fit_steps(module: BaseSuNeRFModule, batches, gradient_clip_val=0.5):
    """Minimal trainer loop with the semantics run_emission.py configures on the Lightning Trainer
    (run_emission.py:65-75): backward, clip_grad_norm_(0.5), Adam step, on_train_batch_end."""
    (optimizer,), _ = module.configure_optimizers()
    optimizer.max_norm = gradient_clip_val          # clip fused into the optimiser step (norm and coefficient stay on device)
    # Early all-reduce of the fine model's slice while the coarse backward still runs: one backward per model and step here, so
    # it is legal -- but it has only ever run with RCCL at world size 1 (no multi-GPU lease so far), so it stays opt-in
    # (SUNERF_OVERLAP=1) until a 2..8-GPU record exists; bench.py follows the same switch.
    optimizer.overlap = env_flag('SUNERF_OVERLAP')
    losses = []
    for i, batch in enumerate(batches):
        optimizer.zero_grad()
        loss = module.training_step(batch, i)
        loss.backward()
        stats = getattr(module, 'last_stats', None)
        optimizer.step(skip_if_positive=None if stats is None else stats[5:6])
        module.on_train_batch_end()     # (takes the all-rank finite check under data parallelism)
        losses.append(loss.detach())
    return losses
