# Board power and shader clock (rocm-smi, read-only) sampled every 0.5 s while the headline kernel runs short (mostly
# zeros in the grid) and long (grid filled) shots: is the data-dependent rate a power / clock effect?
# Run on the GPU box:  bash tools/power_probe.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/power; mkdir -p $O
probe() {  # name, nt, steps
  ( while true; do rocm-smi --showpower --showclocks --json 2>/dev/null | head -c 1500 >> $O/$1.smi; echo >> $O/$1.smi; sleep 0.5; done ) &
  local pid=$!
  python $R/bench.py --leg headline --nt $2 --steps $3 --warmup 2 > $O/$1.json 2> $O/$1.err
  kill $pid
}
probe short 200 60
probe long 3000 4
python3 - <<PY
import json, re
for name in ("short", "long"):
    v = json.load(open("$O/%s.json" % name))["value"]
    pw, ck = [], []
    for line in open("$O/%s.smi" % name):
        line = line.strip()
        if not line.startswith("{"): continue
        try: d = json.loads(line)
        except Exception: continue
        c = d.get("card0", {})
        for k, val in c.items():
            if "Power" in k and "W" in k:
                try: pw.append(float(val))
                except Exception: pass
            if "sclk" in k.lower():
                m = re.search(r"(\d+)\s*Mhz", str(val), re.I)
                if m: ck.append(int(m.group(1)))
    pw = [p for p in pw if p > 0]
    top = sorted(pw)[len(pw) // 2:] if pw else [0]
    print("%-5s %.1f Gpts/s; power samples %d, upper-half mean %.0f W, max %.0f W; sclk samples %s" % (
        name, v, len(pw), sum(top) / len(top), max(pw or [0]), sorted(set(ck))[-5:]))
PY
