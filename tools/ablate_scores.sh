for ab in 0 1 8 9; do
echo "ablate=$ab"; MARIUS_ABLATE=$ab python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('  scores', j['kernels']['lp_scores']['avg_ms'])"
done
