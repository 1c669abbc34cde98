cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/null
for t in none pwconv pwgrad bn_ nhwc_bn conv_ dwconv_lk tapsum "pwconv,pwgrad,tapsum" "bn_,nhwc_bn"; do
  PPEA_NULL=$t timeout -k 10 150 python tools/null_ablation.py > gpurun_out/null/$t.log 2> gpurun_out/null/$t.err
  echo "$t rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/null/$t.log)" >> gpurun_out/null/summary.txt
done
