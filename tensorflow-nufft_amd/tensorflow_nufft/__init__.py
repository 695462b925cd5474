"""TensorFlow NUFFT -- MI355X (gfx950) build of the hot path.

Drop-in for the reference package's public names
(tensorflow_nufft/__init__.py:17-20): `nufft`, `nudft`, `interp`, `spread`,
`Options` and the option enums. Usage: `import tensorflow_nufft as tfft`.
"""
from tensorflow_nufft._lib import InvalidArgumentError
from tensorflow_nufft.nufft_ops import interp, nudft, nufft, spread
from tensorflow_nufft.nufft_options import (DebuggingOptions, FftwOptions,
                                            FftwPlanningRigor, Options, PointsRange)
from tensorflow_nufft.plan import Plan

__version__ = '0.12.0+mi355x.1'
