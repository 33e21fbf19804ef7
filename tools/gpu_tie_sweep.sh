# tie-scan geometry sweep: rebuilds the library on the box per variant (SWT_EXTRA_FLAGS), FastBPE.train on both S85k corpora
export TMPDIR=/tmp
for v in "${@}"; do
  b=${v%%:*}; w=${v##*:}
  export SWT_EXTRA_FLAGS="-DSWT_TIE_BLOCKS=$b -DSWT_TIE_WORDS=$w"
  python -c "import importlib; importlib.import_module('subword-tokenizers_amd._build').build()" || exit 1
  for c in open lex; do
    timeout -k 10 300 python bench.py --workload bpe_train --corpus $c > gpurun_out/sweep_$b_$w_$c.json 2> gpurun_out/sweep.err || { tail -5 gpurun_out/sweep.err; exit 1; }
    python - <<PY
import json
d=json.load(open("gpurun_out/sweep_$b_$w_$c.json"))
print("blocks $b words $w $c:", d["value"], "s/1k  merge_us", d["roofline"]["kernel_us"], flush=True)
PY
  done
done
