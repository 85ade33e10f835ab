"""Rank body of the control-plane tests: every collective of rendezvous.Rendezvous, no torch in the process.
Launched directly (RANK / WORLD_SIZE / MASTER_* in the environment) or under ``python -m torch.distributed.run``
(whose agent holds MASTER_PORT itself -- the case the +1.. port probing exists for)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from full_waveform_inversion_amd.rendezvous import Rendezvous  # noqa: E402


def main():
    out = sys.argv[1]
    with Rendezvous.from_env(timeout=60) as r:
        uid = r.broadcast(bytes(range(128)) if r.rank == 0 else None)
        assert uid == bytes(range(128))
        r.barrier()
        s = r.allreduce([r.rank + 1.0, 2.0])
        assert s == [r.world * (r.world + 1) / 2.0, 2.0 * r.world], s
        assert r.allreduce([float(r.rank)], "max") == [r.world - 1.0]
        assert r.allreduce([float(r.rank)], "min") == [0.0]
        import numpy as np
        a = r.allreduce_array(np.full((3, 2), r.rank + 0.5))
        assert a.shape == (3, 2) and np.all(a == sum(k + 0.5 for k in range(r.world)))
        rows = r.gather(b"rank%d" % r.rank)
        assert (rows == [b"rank%d" % k for k in range(r.world)]) if r.rank == 0 else rows is None
        r.barrier()
        assert "torch" not in sys.modules, "the control plane must not pull torch into the process"
        open("%s.rank%d" % (out, r.rank), "w").write("ok %d %d" % (r.rank, r.world))


if __name__ == "__main__":
    main()
