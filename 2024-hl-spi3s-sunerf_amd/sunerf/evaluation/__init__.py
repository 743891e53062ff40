
# Make this a *portion* of the `sunerf` package: sub-modules that are not mirrored here (data loaders, evaluation,
# run_emission, ...) keep resolving from a reference checkout placed LATER on sys.path, while the mirrored hot-path
# modules resolve from this directory first.
from pkgutil import extend_path
__path__ = extend_path(__path__, __name__)
