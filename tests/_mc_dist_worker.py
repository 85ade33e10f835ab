"""Rank body of the world_size-2 Monte Carlo sharding test.  ``backend`` = "oracle": the device call is
replaced by the CPU oracle (CPU suite); "gpu": the real fwi_mc_invert on device 0 (GPU suite)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from full_waveform_inversion_amd import samplers, source_inversion as si  # noqa: E402
from full_waveform_inversion_amd.rendezvous import Rendezvous  # noqa: E402
from oracle import mc_oracle as mo  # noqa: E402


def oracle_invert(d, G, count, typ, seed, first, amp, metric, norm, allat, return_samples=True, device=0,
                  return_timing=False):
    """Same contract as si.invert_on_device, computed by the oracle (deviates) + the reference maps."""
    M, frac = samplers.from_deviates(typ, mo.device_sampler_deviates(typ, seed, first, count))
    M = M * amp
    sim, like = mo.score_samples(G, d, M, metric, norm, allat)
    return M, (frac if frac is not None else np.zeros(count)), sim, like, mo.posterior(like)


def problem():
    rng = np.random.default_rng(21)
    k, n, t = 4, 9, 48
    G = rng.standard_normal((k, n, t))
    d = np.einsum("kjt,j->kt", G, rng.standard_normal(n)) + 0.1 * rng.standard_normal((k, t))
    return d, G, 1001, "DC_single_force_couple"   # odd count: uneven split 501 + 500


def main():
    out, backend = sys.argv[1], sys.argv[2]
    rdzv = Rendezvous.from_env(timeout=120)
    rank, world = rdzv.rank, rdzv.world
    if backend == "oracle":
        si.invert_on_device = oracle_invert

    def sum_over_ranks(x):
        return rdzv.allreduce([x])[0]

    d, G, N, typ = problem()
    first, M, post, like = si.perform_monte_carlo_sampled_waveform_inversion_sharded(
        d, G, N, rank, world, sum_over_ranks, M_amplitude=1.5, inversion_type=typ, comparison_metric="VR",
        perform_normallised_waveform_inversion=False, compare_all_waveforms_simultaneously=False, seed=9)
    np.savez(out + ".rank%d.npz" % rank, first=first, M=M, post=post, like=like)
    rdzv.barrier()
    rdzv.close()


if __name__ == "__main__":
    main()
