"""Serial restatement of the reference's PRBS generator.  TEST INFRASTRUCTURE.

Reference: ``opticomlib/devices.py:144-182`` -- a Fibonacci LFSR of ``order`` bits: every step emits the
state's bit 0, forms ``new = bit[tap1] ^ bit[tap2]`` (taps of the table at ``:144-152``, zero-based) and
shifts it in from the right.  ``seed`` is taken modulo ``2**order`` (default all ones); 0 becomes 1 with a
UserWarning.  Integer arithmetic only: parity is bit-exact.

Parity status: PINNED by the literal vectors of the reference's own test (``tests/devices_test.py:52-71``,
restated in ``tests/test_oracle_golden.py``) and by ``tests/golden/prbs*.npz`` captured from an import.
Only ``tests/`` may import this module.
"""
from __future__ import annotations

import numpy as np

TAPS = {7: (7, 6), 9: (9, 5), 11: (11, 9), 15: (15, 14), 20: (20, 3), 23: (23, 18), 31: (31, 28)}


def prbs(order: int, length: int | None = None, seed: int | None = None):
    """Returns ``(bits uint8 (length,), final LFSR state)``."""
    if order not in TAPS:
        raise ValueError("The parameter `order` must be one of the following values (7, 9, 11, 15, 20, 23, 31).")
    mask = (1 << order) - 1
    lfsr = seed % (1 << order) if seed is not None else mask
    if lfsr == 0:
        lfsr = 1
    if length is None:
        length = mask
    t1, t2 = TAPS[order][0] - 1, TAPS[order][1] - 1
    out = np.empty(length, dtype=np.uint8)
    for i in range(length):
        out[i] = lfsr & 1
        new = ((lfsr >> t1) ^ (lfsr >> t2)) & 1
        lfsr = ((lfsr << 1) | new) & mask
    return out, lfsr
