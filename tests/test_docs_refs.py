"""The evidence files the documents cite exist (every `profiles/<name>` path in DESIGN.md / DESIGN_NOTES.md / README.md / INTEGRATION.md, every
r03_* / r04_* file name listed in profiles/README.md), and the figures the documents quote about the current round's profiles are the ones a
script generates from the JSON / CSV files (hand-copied per-layer figures had gone stale)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cited_profile_files_exist():
    missing = []
    for doc in ("DESIGN.md", "DESIGN_NOTES.md", "README.md", "INTEGRATION.md"):
        text = open(os.path.join(ROOT, doc)).read()
        for name in set(re.findall(r"profiles/([A-Za-z0-9_.\-]+\.(?:jsonl|json|csv|log|txt))", text)):
            if "*" in name or not os.path.exists(os.path.join(ROOT, "profiles", name)):
                missing.append((doc, name))
    text = open(os.path.join(ROOT, "profiles", "README.md")).read()
    for name in set(re.findall(r"`(r0[345]_[A-Za-z0-9_.\-]+\.(?:jsonl|json|csv|log|txt|md))`", text)):
        if not os.path.exists(os.path.join(ROOT, "profiles", name)):
            missing.append(("profiles/README.md", name))
    assert not missing, missing


def test_cited_test_and_tool_files_exist():
    missing = []
    for doc in ("DESIGN.md", "DESIGN_NOTES.md", "README.md"):
        text = open(os.path.join(ROOT, doc)).read()
        for path in set(re.findall(r"`((?:tests|tools)/[A-Za-z0-9_./\-]+\.(?:py|sh))", text)):
            if not os.path.exists(os.path.join(ROOT, path)):
                missing.append((doc, path))
    assert not missing, missing


def test_quoted_profile_figures_are_the_generated_ones():
    """profiles/r04_summary.md is what tools/summarize_profiles.py makes of the committed JSON / CSV files NOW, and DESIGN.md's block between
    the `generated` markers is that text: a re-collected profile cannot leave older figures behind in the prose."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import summarize_profiles as SP
    tag = "r06"
    fresh = SP.summary(tag)
    path = os.path.join(ROOT, "profiles", "%s_summary.md" % tag)
    assert os.path.exists(path), "run: python tools/summarize_profiles.py %s" % tag
    assert open(path).read() == fresh, "profiles/%s_summary.md is stale: run python tools/summarize_profiles.py %s" % (tag, tag)
    block = SP.design_block(tag)
    assert block is not None, "DESIGN.md lost its generated block"
    assert block == SP.demote(fresh), "DESIGN.md's generated block differs from profiles/%s_summary.md: run python tools/summarize_profiles.py %s" % (tag, tag)
    # and the per-layer table really is in there
    assert "conv3d_14" in block and "Sum over the table" in block
