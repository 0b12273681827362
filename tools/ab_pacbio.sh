# usage (GPU box): bash tools/ab_pacbio.sh [reads] -- configs[3] (-pacbio, 7 kb reads at 15 % error, hg38-sized index): fragment pairs aligned by the device vs planned on the host
cd $GRAFT_REPO_ROOT
N=${1:-200000}
python3 tools/run_configs.py $N 0 > gpurun_out/pb_dev.json 2> gpurun_out/pb_dev.err
KART_AMD_HOST_FRAGMENTS=1 python3 tools/run_configs.py $N 0 > gpurun_out/pb_host.json 2> gpurun_out/pb_host.err
python3 - <<PY
import json
for tag in ("dev", "host"):
    d = json.load(open("gpurun_out/pb_%s.json" % tag))["configs[3] -pacbio"]
    k = d["kart_amd"]
    print(tag, "mapping_seconds", k.get("mapping_seconds"), "reads/s", k.get("reads_per_s_mapping_phase"), "| same records as reference:", d.get("same_records_as_reference"), "| prefix identical to -t 1:", d.get("prefix_identity_vs_reference_t1", {}).get("identical"))
    for l in k.get("log", []):
        if l.startswith(("worker", "fragment", "cpu seconds", "stage seconds")): print("   ", l[:300])
PY
