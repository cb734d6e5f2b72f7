"""The recorded full `pytest -m gpu` run must belong to the kernel sources of THIS tree (a round shipped nine kernel commits
behind its last recorded GPU suite run).  profiles/r04_gputest_final.log is written by tools/gputest_stamp.sh on the GPU box; its first
line carries the sha256 over csrc/*.hip + common.h + the public header (bench.kernel_source_hash)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOG = os.path.join(ROOT, "profiles", "r04_gputest_final.log")


def test_recorded_gpu_suite_run_matches_the_kernel_sources():
    import bench
    assert os.path.exists(LOG), "no recorded GPU suite run: gpurun -- 'bash tools/gputest_stamp.sh r04', then copy the log to profiles/"
    text = open(LOG).read()
    m = re.search(r"kernel_source_hash=([0-9a-f]+)", text.splitlines()[0])
    assert m, "unstamped log"
    assert m.group(1) == bench.kernel_source_hash(), (
        "kernel sources changed after the last recorded GPU suite run (%s != %s): re-run tools/gputest_stamp.sh" % (m.group(1), bench.kernel_source_hash()))
    assert re.search(r"^# rc=0$", text, re.M), "the recorded run was not green"
    tally = re.search(r"(\d+) passed", text)
    assert tally and int(tally.group(1)) >= 290 and " failed" not in text.split("slowest")[-1]


def test_recorded_run_used_the_library_this_tree_builds():
    """The build is bit-reproducible (csrc/Makefile: fixed -cuid per file), `make` records the library's sha256 in the TRACKED file
    lib/libfmri_hip.so.sha256, and tools/gputest_stamp.sh stamps the sha of the .so the suite actually loaded: all three must agree
    (the shipped .so had been rebuilt after the stamped run and nothing would have noticed a difference)."""
    sha_file = os.path.join(ROOT, "fetal-mri-segmentation_amd", "lib", "libfmri_hip.so.sha256")
    assert os.path.exists(sha_file), "build() has not recorded the library's sha256"
    recorded = open(sha_file).read().strip()
    assert re.fullmatch(r"[0-9a-f]{64}", recorded)
    m = re.search(r"libfmri_hip_so_sha256_16=([0-9a-f]{16})", open(LOG).read().splitlines()[0])
    assert m and recorded.startswith(m.group(1)), "the recorded GPU suite ran on another build of the library (%s vs %s)" % (m and m.group(1), recorded[:16])
    so = os.path.join(ROOT, "fetal-mri-segmentation_amd", "lib", "libfmri_hip.so")
    if os.path.exists(so):
        import hashlib
        assert hashlib.sha256(open(so, "rb").read()).hexdigest() == recorded, "libfmri_hip.so in the tree is not the recorded build: run make"
