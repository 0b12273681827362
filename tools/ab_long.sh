# usage (GPU box): bash tools/ab_long.sh [reads] -- configs[3] (-pacbio, 7 kb reads at 15 % error, hg38-sized index): the per-read report on the
# device (kg_longread_batch) at several batch sizes against the host's report (KART_AMD_HOST_LONG=1, round 4), on the same files;
# then KART_AMD_CHECK_ALIGN over CHECK reads (every device record against the host's text)
cd $GRAFT_REPO_ROOT
N=${1:-400000}; CHECK=${CHECK:-200000}
RUN_CONFIGS_NO_REF=1 RUN_CONFIGS_KEEP_INPUTS=1 python3 tools/run_configs.py $N 0 > gpurun_out/ab_long_first.json 2> gpurun_out/ab_long_first.err
CMD=$(python3 -c "import json; d=json.load(open('gpurun_out/ab_long_first.json')); print(' '.join(d['configs[3] -pacbio']['command']))")
echo "command: $CMD"
show() { grep -E "^mapping seconds|^cpu seconds|^device report: [0-9]|^worker thread|^stage seconds|long-read report" $1 | cut -c1-420 | sed 's/^/    /'; grep -o "long-read report on the device[^|]*" $1 | head -1 | sed 's/^/    /'; grep -E "^kg_longread_batch|^KG_LONG_DEBUG_JOBS" $1 | tail -4 | cut -c1-600 | sed 's/^/    /'; }
run() { echo "== $1"; timeout ${RUN_LIMIT:-300} env KART_AMD_VERBOSE=1 $1 $CMD > /tmp/ab_long.log 2>&1; show /tmp/ab_long.log; }
for c in ${CHUNKS:-2048:2048 4096:4096 8192:8192}; do
  run "KART_AMD_PACBIO_CHUNKS=${c%:*} KART_AMD_PACBIO_MAX_CHUNKS=${c#*:}"
done
IFS=";"; for v in ${VARIANTS:-}; do unset IFS; [ -n "$v" ] && run "$v"; IFS=";"; done; unset IFS
[ -z "$NO_HOST_LONG" ] && run "KART_AMD_HOST_LONG=1"
if [ "$CHECK" -gt 0 ]; then
  FQ=$(echo $CMD | sed 's/.*-f \([^ ]*\) .*/\1/')
  head -n $((4 * CHECK)) $FQ > /tmp/ab_long_check.fq
  echo "== KART_AMD_CHECK_ALIGN=1 on $CHECK reads"
  timeout 600 env KART_AMD_VERBOSE=1 KART_AMD_CHECK_ALIGN=1 $(echo $CMD | sed "s#-f $FQ#-f /tmp/ab_long_check.fq#") > /tmp/ab_long_chk.log 2>&1
  grep -E "^CHECK_ALIGN|^device report: [0-9]|^mapping seconds" /tmp/ab_long_chk.log | cut -c1-300 | head -20
  rm -f /tmp/ab_long_check.fq
fi
rm -f $FQ
