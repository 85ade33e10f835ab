#!/usr/bin/env python3
"""hipGraph capture of the step loop against plain stream launches (SURVEY s.7.2).

For each workload: the same forward (and store / adjoint) sweeps with launch_mode = stream and = graph on ONE context
(fwi_set_launch_mode between sweeps, alternating which mode goes first); per mode the device time of the loop (HIP events around it), the host time spent submitting
it, and the share of that spent capturing + instantiating the graph.  One JSON line per workload.
    python tools/graph_probe.py [--rounds 5] > profiles/r04_graph_probe.jsonl
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from full_waveform_inversion_amd import Engine, workloads  # noqa: E402


def probe(name, w, rounds, sweeps=("forward",), **kw):
    """ONE context; the launch mode is switched between sweeps (fwi_set_launch_mode): same buffers, same cache state.
    Round r runs the modes in the order (stream, graph) when r is even and (graph, stream) when odd, so that neither
    mode always follows the other; a first round is discarded."""
    wav = w.wavelet(np.float32)
    out = {"workload": name, "shape": list(w.shape), "nt": w.nt, "engine": {k: str(v) for k, v in kw.items()}}
    res = {m: {s: {"loop_ms": [], "submit_ms": [], "graph_build_ms": []} for s in sweeps} for m in ("stream", "graph")}
    d = {}
    with Engine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml, launch_mode="stream", **kw) as e:
        e.set_model(w.c.astype(np.float32))
        for r in range(rounds + 1):
            for m in (("stream", "graph") if r % 2 == 0 else ("graph", "stream")):
                e.set_launch_mode(m)
                for s in sweeps:
                    if s == "forward":
                        d[m] = e.forward(None, (w.src_idx[:1], wav), w.rec_idx, save=False)
                    elif s == "store":
                        d[m] = e.forward(None, (w.src_idx[:1], wav), w.rec_idx, save=True)
                    else:
                        e.adjoint(d[m])
                    if r:
                        sub, gb = e.last_host_ms()
                        res[m][s]["loop_ms"].append(e.last_loop_ms())
                        res[m][s]["submit_ms"].append(sub)
                        res[m][s]["graph_build_ms"].append(gb)
        out["kernel"] = e.kernel_name
        out["identical_seismograms"] = bool(np.array_equal(d["stream"], d["graph"]))
    for s in sweeps:
        row = {}
        for m in res:
            row[m] = {k: round(float(np.median(v)), 4) for k, v in res[m][s].items()}
            row[m]["loop_ms_min_max"] = [round(float(min(res[m][s]["loop_ms"])), 4), round(float(max(res[m][s]["loop_ms"])), 4)]
            row[m]["us_per_step"] = round(1e3 * row[m]["loop_ms"] / w.nt, 4)
        row["graph_over_stream_loop"] = round(row["graph"]["loop_ms"] / row["stream"]["loop_ms"], 4)
        out[s] = row
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    cases = []
    for n in (256, 512):
        def mk(n=n):
            w = workloads.cfg2(n / 1024.0)
            w.nt = 2000
            return w
        cases.append(("2-D %d^2 sponge" % n, mk, ("forward",), {}))
    cases.append(("configs[0]: 2-D 256^2 O(2) 500 steps", lambda: workloads.cfg1(1.0), ("forward",), {}))
    cases.append(("configs[1]: 2-D 1024^2 sponge", lambda: workloads.cfg2(1.0), ("forward", "store", "adjoint"), {}))
    cases.append(("configs[1]: 2-D 1024^2 CPML", lambda: workloads.cfg2(1.0), ("forward", "store", "adjoint"),
                  dict(abc="cpml", pml_alpha_max=np.pi * 15.0)))

    def c3(npml, nt=300):
        def mk():
            w = workloads.cfg4(1.0, npml=npml)
            w.nt = nt
            return w
        return mk
    cases.append(("3-D 256^3 CPML npml 16 (2 launches / step)", c3(16), ("forward", "store", "adjoint"),
                  dict(abc="cpml", pml_alpha_max=np.pi * 10.0)))
    cases.append(("configs[3]: 3-D 256^3 no border (1 launch / step)", c3(0, 1000), ("forward",), {}))
    for name, mk, sweeps, kw in cases:
        if a.only and a.only not in name:
            continue
        print(json.dumps(probe(name, mk(), a.rounds, sweeps, **kw)), flush=True)


if __name__ == "__main__":
    main()
