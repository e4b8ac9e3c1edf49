#!/usr/bin/env python
"""Event-timed chamfer kernels (ha_chamfer_forward / _backward) at fitting-like sizes: pairs/s and the VALU instruction rate they imply.
usage: chamfer_timing.py [clouds n m]   (default 1920 clouds of 6890 predicted vertices against 2048 observed points)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from humor_amd.chamfer import ChamferDistance          # noqa: E402

VALU_INSTR = 256 * 4 * 2.4e9 / 4                      # wave64 VALU instructions per second of the chip (4 SIMDs per CU, 4 cycles per instruction)
INSTR_PER_PAIR = 19 / 4                                # ISA of the steady-state trip (4 candidates): 10 v_pk_add_f32, 6 v_pk_mul_f32, 2 min, 1 cmp
LDS_BYTES = 256 * 128 * 2.4e9                          # LDS return bandwidth: 128 B per clock and CU
LDS_BYTES_PER_PAIR = 3 * 16 / 4                        # three 16-byte reads per lane and 4 candidates


def timed(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    b, n, m = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (1920, 6890, 2048)
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    x1 = torch.randn(b, n, 3, generator=g).to(dev).requires_grad_(True)
    x2 = torch.randn(b, m, 3, generator=g).to(dev).requires_grad_(True)
    cd = ChamferDistance()

    def fwd():
        with torch.no_grad():
            return cd(x1, x2)

    def fwdbwd():
        x1.grad = None
        x2.grad = None
        d1, d2 = cd(x1, x2)[:2]
        (d1.sum() + d2.sum()).backward()
    tf, tb = timed(fwd), timed(fwdbwd)
    pairs = 2.0 * b * n * m                            # both directions
    rate = pairs / (tf * 1e-3)
    print(f'{b} clouds, {n} x {m} points: forward {tf:.3f} ms = {rate / 1e12:.2f} T pairs/s = {rate / 64 * INSTR_PER_PAIR / VALU_INSTR:.2f} of the VALU '
          f'issue rate ({INSTR_PER_PAIR} instructions per pair) and {rate * LDS_BYTES_PER_PAIR / LDS_BYTES:.2f} of the LDS return bandwidth; '
          f'forward + backward {tb:.3f} ms')


if __name__ == '__main__':
    main()
