"""Rank body of the world_size-2 CPU test (launched by test_host_logic.py): the product's own stdlib
rendezvous as the control plane, the oracle-backed engine in place of the GPU."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from _oracle_engine import OracleEngine  # noqa: E402
from full_waveform_inversion_amd import shots as sh, workloads  # noqa: E402
from full_waveform_inversion_amd.rendezvous import Rendezvous  # noqa: E402


def main():
    out = sys.argv[1]
    rdzv = Rendezvous.from_env(timeout=120)
    ex = sh.HostExchange(rdzv)
    w = workloads.cfg3(0.0625, nshots=5)  # 64 x 64, 5 shots: uneven split 3 + 2
    wav = w.wavelet(np.float64)
    shots = [sh.Shot(w.src_idx[i:i + 1], wav, w.rec_idx) for i in range(len(w.src_idx))]
    e = OracleEngine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml)
    sh.model_data(e, w.c, shots, ex)
    mine = sh.partition_shots(len(shots), ex.rank, ex.world)
    assert all((s.d_obs is not None) == (i in mine) for i, s in enumerate(shots))
    J, g = sh.misfit_and_gradient(e, w.c_init, shots, ex)
    np.savez(out + ".rank%d.npz" % ex.rank, J=J, g=g, mine=np.array(mine))
    assert rdzv.allreduce([ex.rank + 1.0], "max")[0] == ex.world and rdzv.allreduce([ex.rank], "min")[0] == 0
    rdzv.barrier()
    rdzv.close()


if __name__ == "__main__":
    main()
