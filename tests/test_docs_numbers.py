"""DESIGN.md 4.0 / 4.2, README.md's headline and INTEGRATION.md 2 quote their figures from profiles/ through tools/doc_numbers.py:
a document that disagrees with the filed profiles fails here (regenerate with `python tools/doc_numbers.py <tag>`)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generated_blocks_agree_with_the_filed_profiles():
    tag = "round4"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "doc_numbers.py"), tag, "--check"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr


def test_every_profile_file_the_documents_cite_exists():
    import re
    missing = []
    for doc in ("DESIGN.md", "README.md", "INTEGRATION.md", os.path.join("profiles", "README.md")):
        txt = open(os.path.join(ROOT, doc)).read()
        for m in re.finditer(r"profiles/(round\d+_[A-Za-z0-9_]+\.(?:txt|json|csv))", txt):
            if not os.path.exists(os.path.join(ROOT, "profiles", m.group(1))):
                missing.append((doc, m.group(1)))
    assert not missing, missing
