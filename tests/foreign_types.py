"""Stand-ins with the LAYOUT of opticomlib's own signal classes (class names, ``.signal`` / ``.noise`` /
``.n_pol``, a module-level ``gv`` with ``fs`` / ``dt`` / ``f0``, a NULL sentinel that is not an ndarray), to test
that the devices of opticomlib_amd accept the reference library's objects and hand back objects of the
caller's class.  Not the reference's code: just enough structure for ``devices._adopt``."""
import numpy as np


class _Null:
    def __repr__(self):
        return "NULL"


NULL = _Null()


class _Grid:
    def __init__(self):
        self.set(16, 10e9)

    def set(self, sps, R):
        self.sps, self.R = sps, R
        self.fs = R * sps
        self.dt = 1 / self.fs
        self.f0 = 299792458.0 / 1550e-9


gv = _Grid()


class electrical_signal:
    def __init__(self, signal, noise=NULL):
        self.signal = np.asarray(signal)
        self.noise = noise if noise is NULL else np.asarray(noise)
        self.execution_time = 0.0


class optical_signal(electrical_signal):
    def __init__(self, signal, noise=NULL, n_pol=None):
        super().__init__(signal, noise)
        self.n_pol = 2 if self.signal.ndim == 2 else 1
