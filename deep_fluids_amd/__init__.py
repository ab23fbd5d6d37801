"""deep_fluids_amd -- MI355X-native (gfx950) velocity-field train step of Deep Fluids.

``ops`` / ``model`` mirror the reference's ``ops.py`` / ``model.py`` call surface;  ``trainer``
reproduces ``Trainer.build_model`` + ``train_`` (trainer.py:136-184, 232-293; trainer3.py:14-63);
all arithmetic runs in ``csrc/libdeepfluids_hip.so`` (C-ABI: ``include/deepfluids_hip.h``).
"""
from . import _lib  # noqa: F401
from . import ops, model  # noqa: F401

__version__ = "0.1.0"
