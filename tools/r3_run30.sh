export TMPDIR=/tmp
o=gpurun_out/r3ab; mkdir -p $o
for c in dense smooth; do
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $o/kt_$c -- python bench.py --content $c --steps 256 --repeats 2 --min-time 0 --no-cpu-baseline --no-parity --no-profile --no-pmc --second-content "" > $o/kt_$c.log 2>&1
python - $o/kt_$c $c <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("k_recon_lf")]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
iv = iv[len(iv)//4: -len(iv)//8]   # steady part
span = iv[-1][1] - iv[0][0]
busy = 0; cur_s, cur_e = iv[0]
for s, e in iv[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
# overlap: time with 2 kernels
ev = sorted([(s, 1) for s, e in iv] + [(e, -1) for s, e in iv]); c = 0; last = ev[0][0]; t2 = 0
for t, d in ev:
    if c >= 2: t2 += t - last
    last = t; c += d
d = [e - s for s, e in iv]
print(sys.argv[2], "launches", len(iv), "span us", span / 1e3, "busy frac", round(busy / span, 4), "two-at-once frac", round(t2 / span, 4), "mean dur us", sum(d) / len(d) / 1e3, "per launch us", span / len(iv) / 1e3)
PY
done
