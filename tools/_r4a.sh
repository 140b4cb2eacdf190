set -o pipefail
mkdir -p gpurun_out/r4a
python -m pytest tests -m gpu -x -q > gpurun_out/r4a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4a/pytest.log
tail -5 gpurun_out/r4a/pytest.log
for n in 512 256 128 64 960 1920 4096; do
  echo "== N=$n runs"; N=$n python tools/microbench.py 2>/dev/null | sed -n 1,5p
  echo "== N=$n band walk (AC_NO_RUNS=1: threshold only differs)"; AC_NO_RUNS=1 N=$n python tools/microbench.py 2>/dev/null | sed -n 4,5p
done > gpurun_out/r4a/times.txt 2>&1
cat gpurun_out/r4a/times.txt
