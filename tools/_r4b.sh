mkdir -p gpurun_out/r4b
python -m pytest tests -m gpu -q -x -k "fused or golden or short_frames or fuzz or streaming or masking_model or psy_random or bfloat16 or pcm16" > gpurun_out/r4b/pytest.log 2>&1; tail -3 gpurun_out/r4b/pytest.log
for n in 512 256 128 64 960 1920 4096; do echo "== N=$n"; N=$n python tools/microbench.py 2>/dev/null | sed -n '1p;5p'; done
