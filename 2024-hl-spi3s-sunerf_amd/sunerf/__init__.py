"""Drop-in ``sunerf`` package for the hot path of FrontierDevelopmentLab/2024-HL-SPI3S-SuNeRF on MI355X.

Same import paths, class names, constructor signatures, output-dict keys and state-dict keys as the reference's
``sunerf.train.sampling``, ``sunerf.model.model``, ``sunerf.rendering.{base_tracing,emission}``,
``sunerf.train.scaling`` and ``sunerf.model.sunerf``, so ``run_emission.py`` and reference
checkpoints / ``.snf`` pickles work against it.  All numerics of the render path run in the HIP kernels of
``../csrc`` through the C ABI ``include/sunerf_hip.h``; there is no CPU implementation in this package.
"""

# Make this a *portion* of the `sunerf` package: sub-modules that are not mirrored here (data loaders, evaluation,
# run_emission, ...) keep resolving from a reference checkout placed LATER on sys.path, while the mirrored hot-path
# modules resolve from this directory first.
from pkgutil import extend_path
__path__ = extend_path(__path__, __name__)
