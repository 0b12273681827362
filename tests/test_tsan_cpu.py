"""CPU: the host pipeline under ThreadSanitizer (tools/tsan_host.py; SURVEY.md section 5 "race detection").  The short form: the CPU
build of the pipeline with -fsanitize=thread over a seeded paired-end set at 2 / 4 / 8 threads, -m, small batches, gz input through the several-thread reader, a sharded run and
the long-read golden -- no report, outputs identical to the plain binary's.  The full job's log is profiles/r06_tsan.log."""
import os
import subprocess
import sys

from conftest import ROOT


def test_host_pipeline_under_thread_sanitizer():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "tsan_host.py"), "--pairs", "12000", "--only", "14"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-3000:]
    assert "14 runs, 0 reports or differences" in out, out[-1500:]
