# Every clock rocm-smi reports (sclk, mclk, fclk, socclk ...) and the board power, sampled every 0.4 s while the headline
# kernel runs a short shot (grid mostly zeros) and a long one (grid filled): which clock, if any, follows the data?
# Run on the GPU box:  bash tools/clock_probe.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/clocks; mkdir -p $O; rm -f $O/*.smi
probe() {  # name, nt, steps
  ( while true; do rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d '\n' >> $O/$1.smi; echo >> $O/$1.smi; sleep 0.4; done ) &
  local pid=$!
  python3 $R/bench.py --leg headline --nt $2 --steps $3 --warmup 2 > $O/$1.json 2> $O/$1.err
  kill $pid
}
probe short 200 90
probe long 3000 6
python3 - <<PY
import json, re, collections
for name in ("short", "long"):
    v = json.load(open("$O/%s.json" % name))["value"]
    vals = collections.defaultdict(list)
    for line in open("$O/%s.smi" % name):
        line = line.strip()
        if not line.startswith("{"): continue
        try: d = json.loads(line)
        except Exception: continue
        for k, val in d.get("card0", {}).items():
            m = re.search(r"([0-9.]+)", str(val))
            if m: vals[k].append(float(m.group(1)))
    print("%s: %.1f Gpts/s" % (name, v))
    for k, x in sorted(vals.items()):
        x = sorted(x)
        print("   %-45s n=%3d  min %8.1f  median %8.1f  max %8.1f" % (k[:45], len(x), x[0], x[len(x) // 2], x[-1]))
PY
