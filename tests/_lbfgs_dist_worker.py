"""Rank body of the world_size-2 OPTIMISER-LOOP test (launched by test_host_logic.py): configs[4]'s pattern -- shots
sharded rank::world, one gradient sum per evaluation, L-BFGS on every rank in lock-step -- on the product's own
stdlib control plane with the oracle-backed engine in place of the GPU.  The reference's analogue is its driver loop
around the forked workers (full_waveform_inversion.py:786-870)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from _oracle_engine import OracleEngine  # noqa: E402
from full_waveform_inversion_amd import shots as sh, workloads  # noqa: E402
from full_waveform_inversion_amd.lbfgs import lbfgs  # noqa: E402
from full_waveform_inversion_amd.rendezvous import Rendezvous  # noqa: E402


def problem():
    w = workloads.cfg3(0.0625, nshots=5)  # 64 x 64, 5 shots: uneven split 3 + 2
    wav = w.wavelet(np.float64)
    shots = [sh.Shot(w.src_idx[i:i + 1], wav, w.rec_idx) for i in range(len(w.src_idx))]
    e = OracleEngine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml)
    return w, shots, e


def run(first_step, ex=None, checkpoint=None):
    w, shots, e = problem()
    sh.model_data(e, w.c, shots, ex)
    return lbfgs(lambda m: sh.misfit_and_gradient(e, m, shots, ex), w.c_init, maxiter=3, history=4,
                 first_step=first_step, bounds=(1000.0, 4000.0), checkpoint=checkpoint)


def main():
    out, first_step = sys.argv[1], float(sys.argv[2])
    rdzv = Rendezvous.from_env(timeout=120)
    ex = sh.HostExchange(rdzv)
    # only rank 0 writes the optimiser state (every rank used to write the same path)
    x, f, log = run(first_step, ex, checkpoint=(out + ".state.npz") if ex.rank == 0 else None)
    np.savez(out + ".rank%d.npz" % ex.rank, x=x, f=f, log=np.frombuffer(json.dumps(log).encode(), np.uint8))
    rdzv.barrier()
    rdzv.close()


if __name__ == "__main__":
    main()
