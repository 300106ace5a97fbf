"""The documents cite measured evidence by file name: every `profiles/...` file that DESIGN.md, README.md, INTEGRATION.md,
profiles/README.md or tools/README.md names must exist in the tree (wildcards and elided names are skipped), and every tool the
tools/ README lists must be there.  Catches a renamed or never-committed evidence file before a reader does."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ["DESIGN.md", "README.md", "INTEGRATION.md", "profiles/README.md", "profiles/HISTORY.md", "tools/README.md"]


def test_cited_evidence_files_exist():
    missing = []
    for doc in DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        for m in sorted(set(re.findall(r"`((?:profiles/)?(?:r[1-9]_|k1_)[A-Za-z0-9_.*…]+\.(?:json|jsonl|csv|txt))`", text))):
            if "*" in m or "…" in m:
                continue
            path = m if m.startswith("profiles/") else "profiles/" + m
            if not os.path.exists(os.path.join(ROOT, path)):
                missing.append((doc, m))
    assert not missing, missing


def test_listed_tools_exist():
    text = open(os.path.join(ROOT, "tools", "README.md")).read()
    names = set()
    for row in text.split("\n"):
        if row.startswith("| `"):
            for m in re.findall(r"`(?:\[[^\]]*\] )?([a-z0-9_]+\.(?:py|sh|hip))", row.split("|")[1]):
                names.add(m)
    assert len(names) > 20
    missing = sorted(n for n in names if not os.path.exists(os.path.join(ROOT, "tools", n)))
    assert not missing, missing
