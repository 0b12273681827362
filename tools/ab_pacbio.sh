# usage (GPU box): bash tools/ab_pacbio.sh [reads] -- configs[3] (-pacbio, 7 kb reads at 15 % error, hg38-sized index): fragment pairs aligned by the device
# (at several batch sizes) vs planned on the host
cd $GRAFT_REPO_ROOT
N=${1:-200000}
show() { python3 - "$1" "$2" <<PY
import json, sys
d = json.load(open(sys.argv[1]))["configs[3] -pacbio"]
k = d["kart_amd"]
print(sys.argv[2], "| mapping_seconds", k.get("mapping_seconds"), "reads/s", k.get("reads_per_s_mapping_phase"), "| same records as reference:", d.get("same_records_as_reference"), "| prefix identical to -t 1:", d.get("prefix_identity_vs_reference_t1", {}).get("identical"))
for l in k.get("log", []):
    if l.startswith(("worker", "fragment", "cpu seconds", "stage seconds")): print("   ", l[:260])
    if "fragment service:" in l: print("   ", l[l.index("fragment service:"):][:160])
PY
}
for c in ${CHUNKS:-1024:1024 2048:2048 4096:4096}; do
  RUN_CONFIGS_NO_REF=1 KART_AMD_PACBIO_CHUNKS=${c%:*} KART_AMD_PACBIO_MAX_CHUNKS=${c#*:} python3 tools/run_configs.py $N 0 > gpurun_out/pb_dev_$c.json 2> gpurun_out/pb_dev.err; show gpurun_out/pb_dev_$c.json "device fragments, batches of ${c%:*} chunks (of 10 reads) doubling up to ${c#*:}"
done
RUN_CONFIGS_NO_REF=1 KART_AMD_HOST_FRAGMENTS=1 python3 tools/run_configs.py $N 0 > gpurun_out/pb_host.json 2> gpurun_out/pb_host.err; show gpurun_out/pb_host.json "host planning (round 2)"
