#!/usr/bin/env python3
"""tools/compose_design.py -- one-off of round 4: DESIGN.md re-ordered as VERDICT r03 asked (current state and current numbers
first, one figure per quantity; the record of rounds 1-3 and the measured-and-rejected variants in appendices).  Reads the
round-3 DESIGN.md from git (HEAD~N given on the command line, default the round-3 commit) and the hand-written front part
from tools/design_front.md; writes DESIGN.md."""
import re
import subprocess
import sys

rev = sys.argv[1] if len(sys.argv) > 1 else "766412b"
old = subprocess.run(["git", "show", f"{rev}:DESIGN.md"], capture_output=True, text=True, check=True).stdout
heads = [(m.start(), m.group(0)) for m in re.finditer(r"^## .*$|^### .*$", old, flags=re.M)]


def section(prefix, level="## "):
    """text of the section whose heading starts with prefix, up to the next heading of the same or a higher level"""
    for k, (pos, h) in enumerate(heads):
        if h.startswith(level + prefix):
            end = len(old)
            for pos2, h2 in heads[k + 1:]:
                if h2.startswith("## ") or (level == "### " and h2.startswith("### ")):
                    end = pos2
                    break
            return old[pos:end].rstrip() + "\n"
    raise KeyError(prefix)


def body(text):
    """a section without its heading line"""
    return text.split("\n", 1)[1].lstrip("\n")


front = open("tools/design_front.md").read()
parts = {
    "HIT_DEFINITION": body(section("3. Hit definition")),
    "OPTIONS_N4": body(section("7b. Opt-in options")),
    "DIVERGENCES": body(section("8. Deliberate divergences")),
    "NN_KERNEL": body(section("4.3 ", "### ")),
    "STATS_KERNEL_R2": body(section("4.5 ", "### ")),
    "RNG_R3": body(section("4.6 ", "### ")),
    "BUILD_R3": body(section("2b. The scene build on the GPU")),
    "ORACLE_R3": body(section("7. Oracle and parity")),
}
# wording of the re-used round-2/3 text where it would read as a current figure
parts["BUILD_R3"] = parts["BUILD_R3"].replace(
    "*Cost* (MI355X, `tools/build_compare.py`, `profiles/r03_build_compare.txt`; mesh in pageable host memory, steady state):",
    "*Cost as measured in ROUND 3* (level by level to the bottom; `profiles/r03_build_compare.txt`; the current figures are in "
    "section 1 and in the round-4 paragraph below):")
parts["STATS_KERNEL_R2"] = parts["STATS_KERNEL_R2"].replace(
    "One 256-thread workgroup per frame:", "Round 2's form was one 256-thread workgroup per frame (its chunk reduction is what round 4 keeps):")
parts["STATS_KERNEL_R2"] = parts["STATS_KERNEL_R2"].replace(
    "HBM-bound in principle (two passes over 4 B per value); it runs on\nits own stream beside the next chunk's trace and is not visible in the call time.",
    "HBM-bound in principle (two passes over 4 B per value).")
renumber = [("section 4.1", "appendix B"), ("§4.1", "appendix B"), ("section 2b", "section 5.4"), ("§2b", "section 5.4"),
            ("section 4.6", "section 5.8"), ("§4.6", "section 5.8"), ("§4.5", "section 5.6"), ("section 4.5", "section 5.6"),
            ("(§5)", "(appendix A.3)"), ("§5;", "appendix A.3;"), ("(§3)", "(section 4)"), ("§3;", "section 4;"), ("§3)", "section 4)"),
            ("section 5)", "appendix A.3)"), ("DESIGN.md section 5", "DESIGN.md appendix A.3")]
for k in parts:
    for a, b in renumber:
        parts[k] = parts[k].replace(a, b)
for k, v in parts.items():
    front = front.replace("{{" + k + "}}", v.rstrip() + "\n")
appendix = "\n".join([
    "# Appendices -- the record of rounds 1-3, as written then\n\nSuperseded figures below are NOT current (section 1 is).  Section numbers inside the appendices are the round-3 file's: "
    "4.1 = appendix B, 4.4 and 6 = appendix C, 5 = A.3, 2b = section 5.4, 4.5 / 4.6 = sections 5.6 / 5.8, 7 = section 7.\n",
    "# Appendix A -- round by round\n",
    "## A.1 Round 3 against VERDICT r02, item by item\n", body(section("0a. Round 3 against VERDICT r02")),
    "## A.2 Round 2 against VERDICT r01, item by item\n", body(section("0b. Round 2 against VERDICT r01")),
    "## A.3 Measurements of rounds 2 and 3 (one box per file; boxes of the pool differ by up to 10 %)\n", body(section("5. Measurements")),
    "# Appendix B -- the trace kernel: how it got here, and what was built, measured and rejected (rounds 1-3)\n",
    body(section("4.1 ", "### ")),
    "# Appendix C -- the cloud rebuild kernel and the multi-GPU step, rounds 1-3\n",
    body(section("4.4 ", "### ")), body(section("6. Multi-GPU")),
])
open("DESIGN.md", "w").write(front.rstrip() + "\n\n---\n\n" + appendix)
print("DESIGN.md:", len(open("DESIGN.md").read().splitlines()), "lines")
