for ab in 0 2 4 8 16 12 6 10 14; do
echo "ablate=$ab"; MARIUS_KERNELS=res MARIUS_ABLATE=$ab python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('  grad_adj', j['kernels']['lp_grad_adj']['avg_ms'])"
done
