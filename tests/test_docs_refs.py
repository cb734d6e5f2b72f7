"""The evidence files the documents cite exist: every `profiles/<name>` path in DESIGN.md / README.md / INTEGRATION.md, and every r03_* file
name listed in profiles/README.md."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cited_profile_files_exist():
    missing = []
    for doc in ("DESIGN.md", "README.md", "INTEGRATION.md"):
        text = open(os.path.join(ROOT, doc)).read()
        for name in set(re.findall(r"profiles/([A-Za-z0-9_.\-]+\.(?:jsonl|json|csv|log|txt))", text)):
            if "*" in name or not os.path.exists(os.path.join(ROOT, "profiles", name)):
                missing.append((doc, name))
    text = open(os.path.join(ROOT, "profiles", "README.md")).read()
    for name in set(re.findall(r"`(r03_[A-Za-z0-9_.\-]+\.(?:jsonl|json|csv|log|txt))`", text)):
        if not os.path.exists(os.path.join(ROOT, "profiles", name)):
            missing.append(("profiles/README.md", name))
    assert not missing, missing


def test_cited_test_and_tool_files_exist():
    missing = []
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    for path in set(re.findall(r"`((?:tests|tools)/[A-Za-z0-9_./\-]+\.(?:py|sh))", text)):
        if not os.path.exists(os.path.join(ROOT, path)):
            missing.append(path)
    assert not missing, missing
