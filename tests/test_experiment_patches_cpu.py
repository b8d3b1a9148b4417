"""The timing-ablation switches and the feature pricing live OUTSIDE the product sources, as patches under experiments/ (applied to copies
by tools/ablate_*.sh, tools/ab_c5_build.sh, tools/price_w64.sh).  A patch that no longer applies would silently take the measurement recipes of
profiles/ away: every one of them must apply to the tree as it stands -- and the product kernel files must carry none of the switches themselves."""
import glob
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCHES = sorted(glob.glob(os.path.join(ROOT, "experiments", "ablations", "*.patch"))) + [os.path.join(ROOT, "experiments", "fa_fwd_w64_price.patch")]


@pytest.mark.parametrize("patch", PATCHES, ids=lambda p: os.path.relpath(p, ROOT))
def test_patch_applies_to_the_tree(patch):
    r = subprocess.run(["patch", "-p1", "--dry-run", "--batch", "-i", patch], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0 and "FAILED" not in r.stdout and "fuzz" not in r.stdout, r.stdout + r.stderr


def test_product_kernels_carry_no_ablation_switches():
    csrc = os.path.join(ROOT, "flash-attention_amd", "csrc")
    bad = []
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h", ".cpp")):
            for n, line in enumerate(open(os.path.join(csrc, f)), 1):
                code = line.split("//")[0]
                if re.search(r"\bFA_\w*_ABL\b|\bFA_ABL\b|\bFA_EXPERIMENTS\b|\bFA_FZ_STATS\b|\bFA_IL_EXPERIMENTS\b", code):
                    bad.append("%s:%d: %s" % (f, n, line.strip()[:120]))
    assert not bad, "\n".join(bad)
