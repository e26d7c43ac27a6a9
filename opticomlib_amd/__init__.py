"""opticomlib_amd -- the split-step Fourier fibre path of opticomlib on AMD Instinct MI355X.

Drop-in for ``opticomlib.devices.FIBER`` / ``DBP`` / ``DM``, the zero-phase filters ``LPF`` / ``BPF`` and
the receiver front-end ``PD`` / ``EDFA`` (and the slice of ``optical_signal`` / ``electrical_signal`` /
``gv`` they touch); everything else of opticomlib is out of scope.
The arithmetic runs in hand-written HIP kernels (``csrc/``) behind the C ABI declared in
``include/ssfm_amd.h``.
"""
from .typing import NULL, binary_sequence, electrical_signal, gv, optical_signal
from .devices import BPF, DAC, DBP, DM, EDFA, FIBER, LASER, LPF, MZM, PD, PRBS, device_rng_seed
from ._lib import C64, C128, Plan, SsfmError, device_count

__all__ = ["NULL", "gv", "optical_signal", "electrical_signal", "FIBER", "DBP", "DM", "LPF", "BPF", "PD", "EDFA", "PRBS", "DAC", "LASER", "MZM", "binary_sequence", "device_rng_seed", "Plan", "SsfmError", "device_count", "C64", "C128"]
__version__ = "0.1.0"
