"""CPU: the registry of environment knobs (kart_amd/csrc/host/knobs.inc, printed by `kart-amd -knobs`) against the sources: every
getenv() under kart_amd/csrc names a listed variable, every listed variable is read somewhere, and the README points at the registry."""
import os
import re

from conftest import ROOT

CSRC = os.path.join(ROOT, "kart_amd", "csrc")


def _read_names():
    names = {}
    for d, _, fs in os.walk(CSRC):
        if os.path.basename(d) == "build":
            continue
        for f in fs:
            if not f.endswith((".hip", ".inc", ".cpp", ".hpp", ".h")) or f == "knobs.inc":
                continue
            for m in re.finditer(r'getenv\("([A-Z0-9_]+)"\)', open(os.path.join(d, f), errors="replace").read()):
                names.setdefault(m.group(1), os.path.join(os.path.relpath(d, ROOT), f))
    return names


def test_every_knob_is_registered_and_every_entry_is_read():
    table = re.findall(r'^\t\{"([A-Z0-9_]+)", "([TADM])", "(\w+)", "([^"]+)"\},', open(os.path.join(CSRC, "host", "knobs.inc")).read(), re.M)
    listed = [t[0] for t in table]
    assert len(listed) == len(set(listed)), "an entry twice"
    read = _read_names()
    missing = sorted(set(read) - set(listed))
    stale = sorted(set(listed) - set(read))
    assert not missing, "read by the sources, not in knobs.inc: %s" % ", ".join("%s (%s)" % (n, read[n]) for n in missing)
    assert not stale, "listed in knobs.inc, read nowhere: %s" % ", ".join(stale)
    assert all(len(t[3]) > 10 for t in table)


def test_the_cli_prints_the_registry():
    assert '"-knobs"' in open(os.path.join(CSRC, "host", "cli.cpp")).read()
    assert "knobs" in open(os.path.join(ROOT, "README.md")).read()
