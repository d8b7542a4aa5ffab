for c in 4 8 16 32; do
  echo "chunk $c: $(PT_QUEUE_CHUNK=$c python bench.py --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
done
