"""After a product kernel file changed: carry the experiment patches (experiments/ablations/*.patch, experiments/fa_fwd_w64_price.patch)
over to the new text.  For every file a patch touches: old product text = `git show REV:file` (REV defaults to HEAD), old patched text = that + the patch as committed
at REV, new patched text = a three-way merge (git merge-file) of the working-tree file with the two, new patch = diff(working tree, new patched).  Conflicts are left
in /tmp/rebase_patches/<file>.merged for a hand merge (the script says which); tests/test_experiment_patches_cpu.py checks the result applies.
usage: python tools/rebase_experiment_patches.py [REV]"""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REV = sys.argv[1] if len(sys.argv) > 1 else "HEAD"
TMP = "/tmp/rebase_patches"
PATCHES = sorted(os.path.join("experiments", "ablations", f) for f in os.listdir(os.path.join(ROOT, "experiments", "ablations")) if f.endswith(".patch")) + \
    ["experiments/fa_fwd_w64_price.patch"]


def git_show(path):
    return subprocess.run(["git", "show", f"{REV}:{path}"], cwd=ROOT, capture_output=True, text=True, check=True).stdout


def split_patch(text):
    """{file: its part of the patch}"""
    parts, cur, name = {}, [], None
    for line in text.splitlines(keepends=True):
        m = re.match(r"^--- a/(\S+)", line)
        if m:
            if name: parts[name] = "".join(cur)
            name, cur = m.group(1), []
        cur.append(line)
    if name: parts[name] = "".join(cur)
    return parts


def main():
    os.makedirs(TMP, exist_ok=True)
    bad = []
    for patch in PATCHES:
        old_parts = split_patch(git_show(patch))
        new_text = ""
        for f, part in old_parts.items():
            tag = os.path.join(TMP, os.path.basename(patch) + "." + os.path.basename(f))
            open(tag + ".base", "w").write(git_show(f))
            open(tag + ".sw", "w").write(git_show(f))
            subprocess.run(["patch", "-s", tag + ".sw"], input=part.replace(f"a/{f}", tag + ".sw").replace(f"b/{f}", tag + ".sw"), text=True, check=True)
            cur = open(os.path.join(ROOT, f)).read()
            open(tag + ".merged", "w").write(cur)
            rc = subprocess.run(["git", "merge-file", "-q", tag + ".merged", tag + ".base", tag + ".sw"]).returncode
            if rc != 0:
                bad.append(tag + ".merged")
                continue
            d = subprocess.run(["diff", "-u", "--label", f"a/{f}", "--label", f"b/{f}", os.path.join(ROOT, f), tag + ".merged"], capture_output=True, text=True).stdout
            new_text += d
        if not any(b.startswith(os.path.join(TMP, os.path.basename(patch))) for b in bad):
            open(os.path.join(ROOT, patch), "w").write(new_text)
            print("rebased", patch)
    for b in bad:
        print("CONFLICT: merge by hand, then diff against the product file:", b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
