#!/usr/bin/env python3
"""Regenerate tests/golden/reference_runner_cases_ngpu{4,8}.txt.gz: the case lists the reference's own
tests/test_runner.py derives from its tests/test_config.yaml for 4 and 8 GPUs (the `*_cc` configurations), obtained by
running that runner with a stand-in launcher that keeps the generated `<config>_cases.txt` and launches nothing.
Lines with the managed-memory flag (-m) are dropped: this library has no managed-memory special case, so they would
repeat their neighbours.  Needs /root/reference (development container only); the fixture is what travels.

    python tests/golden/make_runner_cases.py [/root/reference]
"""
import gzip
import os
import stat
import subprocess
import sys
import tempfile

CONFIGS = ["transpose_test_cc", "transpose_test_halo_cc", "transpose_test_padding_cc", "transpose_test_gdimdist_cc",
           "transpose_test_mix_cc", "transpose_test_ac_cc", "transpose_test_rank_order_cc", "halo_test_cc",
           "halo_test_halomix_cc", "halo_test_padding_cc", "halo_test_gdimdist_cc", "halo_test_mix_cc", "halo_test_ac_cc",
           "halo_test_rank_order_cc"]


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    here = os.path.dirname(os.path.abspath(__file__))
    work = tempfile.mkdtemp(prefix="runner_cases_")
    with open(os.path.join(ref, "tests", "test_config.yaml")) as f, open(os.path.join(work, "test_config.yaml"), "w") as g:
        g.write(f.read())
    launcher = os.path.join(work, "launch.sh")
    with open(launcher, "w") as f:
        f.write('#!/bin/bash\nfor a in "$@"; do :; done\ncp "$a" "%s/kept_$(basename "$a")"\necho "Passed all tests."\n' % work)
    os.chmod(launcher, os.stat(launcher).st_mode | stat.S_IEXEC)
    for ngpu in (4, 8):
        generate(ref, work, launcher, here, ngpu, CONFIGS, "reference_runner_cases_ngpu%d.txt.gz" % ngpu)
    # the Fortran flavour of the same sweeps (one-based --ax / --mem_order), for the twins in tests/fortran
    generate(ref, work, launcher, here, 4, [c[:-3] + "_fortran" for c in CONFIGS], "reference_runner_cases_fortran_ngpu4.txt.gz")


def generate(ref, work, launcher, here, ngpu, configs, name):
    out = []
    for cfg in configs:
        subprocess.run([sys.executable, os.path.join(ref, "tests", "test_runner.py"), "--launcher_cmd", launcher, "--ngpu",
                        str(ngpu), cfg], cwd=work, check=True, capture_output=True)
        with open(os.path.join(work, "kept_%s_cases.txt" % cfg)) as f:
            lines = [" ".join(l.split()) for l in f if l.strip()]
        kept = [l for l in lines if "-m" not in l.split()]
        out.append("# config: %s (%d of %d lines, managed-memory variants dropped)" % (cfg, len(kept), len(lines)))
        out.extend(kept)
    path = os.path.join(here, name)
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(("\n".join(out) + "\n").encode())
    print(path, os.path.getsize(path), "bytes,", sum(1 for l in out if not l.startswith("#")), "cases")


if __name__ == "__main__":
    main()
