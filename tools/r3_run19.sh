export TMPDIR=/tmp
o=gpurun_out/r3r; mkdir -p $o
for k in dense typical; do
  THIP_FE_PROF=1 timeout 300 python bench.py --mode e2e --e2e-size 720p --packets $k --no-native --loops 8 2>&1 | grep -v "^{" | tee $o/fe_prof_$k.txt
done
