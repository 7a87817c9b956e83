"""CPU: the working tree that travels to a GPU box must fit.

`gpurun` (and the driver's round-end GPU run) ships /root/repo minus .git/, gpurun_out/ and the paths of .gpurunignore,
and REFUSES a snapshot above 512 MiB -- which would cost every GPU test, smoke() and the bench line at once.  The
git-ignored oracle caches under profiles/_psi_cache/ are the only large files here; since round 4 none of them travels
by default (the GPU tests read committed fixtures under tests/golden/) and the tree is kept under 250 MiB."""
import fnmatch
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIMIT_MIB = 512
MARGIN_MIB = 262         # the tree is kept under 250 MiB: nothing of the oracle caches travels by default


BUILT_IN = (".git/", "gpurun_out/")


def _ignored(rel, patterns):
    """Conservative model of the client's matching, checked against what a call really pushed (321 MiB computed here,
    `push 321 MiB` reported): a `dir/*` or file glob of .gpurunignore is honoured, a bare `dir/` line was NOT (the build
    objects travelled until `brie_amd/build/*` was added)."""
    if any(rel.startswith(p) for p in BUILT_IN):
        return True
    return any(not p.endswith("/") and fnmatch.fnmatch(rel, p) for p in patterns)


def _patterns():
    path = os.path.join(ROOT, ".gpurunignore")
    if not os.path.exists(path):
        return []
    return [l.strip() for l in open(path) if l.strip() and not l.startswith("#")]


def test_snapshot_fits_the_gpu_box_limit():
    pats = _patterns()
    total, big = 0, []
    for d, dirs, files in os.walk(ROOT):
        rel_d = os.path.relpath(d, ROOT)
        rel_d = "" if rel_d == "." else rel_d + "/"
        dirs[:] = [x for x in dirs if not _ignored(rel_d + x + "/", pats)]
        for f in files:
            rel = rel_d + f
            if _ignored(rel, pats):
                continue
            try:
                n = os.path.getsize(os.path.join(d, f))
            except OSError:
                continue
            total += n
            if n > (16 << 20):
                big.append((n >> 20, rel))
    mib = total / float(1 << 20)
    assert mib < LIMIT_MIB - MARGIN_MIB, ("snapshot %.0f MiB: list more of profiles/_psi_cache in .gpurunignore" % mib, sorted(big)[-8:])


def test_what_the_gpu_parity_tests_read_is_in_git_not_in_a_cache():
    """The gene-sample parity tests (tests/test_gpu_fullsize.py::test_psi_null_rule_on_gene_samples...) read committed
    fixtures; the record of the full caches (sha256, size, regeneration command) names a fixture that exists."""
    import json
    rec_path = os.path.join(ROOT, "tests", "golden", "psi_null_caches.json")
    assert os.path.exists(rec_path)
    rec = json.load(open(rec_path))
    assert rec, "no case recorded"
    for case, r in rec.items():
        fx = os.path.join(ROOT, "tests", "golden", r["fixture"])
        assert os.path.exists(fx) and os.path.getsize(fx) < (32 << 20), (case, fx)
        assert "psi_null.py --null" in r["regenerate"] and all(len(f["sha256"]) == 64 for f in r["files"].values())


def test_the_documents_cite_files_that_exist_and_profiles_stays_pruned():
    """`profiles/` keeps what the documents cite (under 200 tracked files), and what they cite by an explicit
    path is in the tree -- a pruned artefact must not leave a dangling citation behind (wildcard / brace patterns and the
    documents of earlier rounds, whose header points at the git history, are not checked)."""
    import re
    import subprocess
    missing = []
    for doc in ("DESIGN.md", "README.md", "BASELINE.md", "SCALE.md", "INTEGRATION.md", os.path.join("docs", "evidence_r5.md"), os.path.join("docs", "evidence_r6.md"),
                os.path.join("profiles", "README.md")):
        text = open(os.path.join(ROOT, doc)).read()
        for m in set(re.findall(r"profiles/[A-Za-z0-9_./{},*<>-]+", text)):
            path = m.rstrip(".,);:")
            if any(ch in path for ch in "{*<") or path.endswith("/"):
                continue
            if not os.path.exists(os.path.join(ROOT, path)):
                missing.append((doc, path))
    assert not missing, missing
    try:
        tracked = subprocess.run(["git", "ls-files", "profiles"], cwd=ROOT, capture_output=True, text=True, check=True).stdout.split()
    except (OSError, subprocess.CalledProcessError):
        return                                                     # not a git checkout (the GPU box's snapshot)
    if tracked:
        assert len(tracked) < 200, len(tracked)
