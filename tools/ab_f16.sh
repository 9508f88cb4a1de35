# bf16 against half-precision query/key operands on one box: same kernels, same MFMA rate
for rep in 1 2; do
for c in bf16 f16; do
  for a in "4096 40 1 200" "1024 80 1 200" "4096 40 4 50"; do echo -n "$c: "; timeout -k 10 60 python3 tools/prof_attn.py $a $c 2>/dev/null | head -1 || exit 1; done
done
done
timeout -k 10 200 python3 bench.py --compute f16 --no-e2e --no-cpu-baseline --steps 20 --warmup 6
timeout -k 10 200 python3 bench.py --compute bf16 --no-e2e --no-cpu-baseline --steps 20 --warmup 6
