"""Find cells that take Lewiner's sub-cases 6.1.2 and 7.4.2 (test infrastructure; writes tests/golden/mc_subcase_cells.json).

Round 5 recorded that 6.1.2 / 7.4.2 / 12.1.2 / 13.5.2 never occurred in 7 M random cells (nor in 192 M heavy-tailed case-6 cells): with
scikit-image's tests a face that reads "separated" does not meet an interior that reads "connected" - EXCEPT THROUGH EXACT TIES.  test_face
computes q = A C - B D on the face and, when |q| < eps, returns `face >= 0` whatever the interior looks like; so a cell whose tested face
has A C == B D exactly (small integers do it) and a negative face id in TEST6 / TEST7 drops into the interior test with the face reading
"separated", and 6.1.2 / 7.4.2 come out.  This script enumerates corner magnitudes in {1, 2, 4} under every sign pattern of cases 6 and 7
with the oracle's own cell classifier and keeps the first 40 cells of each tag.  (The same enumeration with magnitudes {1, 2, 3, 4, 6}
over all 19 M cells of cases 12 and 13 finds no 12.1.2 and no 13.5.2: those stay unexercised, and a penalty-minimising search -
scipy differential evolution over log-magnitudes - ends ON the face-test boundary for every configuration, never across it.)

    python tests/golden/find_mc_subcase_cells.py     then     /opt/conda/bin/python3.9 tests/golden/make_golden_mc.py lewiner_subcases
"""
import itertools
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import marching_cubes as OM  # noqa: E402


def main():
    found = {"6.1.2": [], "7.4.2": []}
    for idx in range(256):
        case = int(OM.L["CASES"][idx][0])
        if case not in (6, 7):
            continue
        signs = [1.0 if (idx >> p) & 1 else -1.0 for p in range(8)]
        for m in itertools.product((1.0, 2.0, 4.0), repeat=8):
            cube = [s * v for s, v in zip(signs, m)]
            tag = OM.cell_triangles(cube)[1]
            if tag in found and len(found[tag]) < 40:
                found[tag].append(cube)
    assert all(len(v) == 40 for v in found.values()), {k: len(v) for k, v in found.items()}
    with open(os.path.join(HERE, "mc_subcase_cells.json"), "w") as fh:
        json.dump(found, fh)
    print({k: len(v) for k, v in found.items()})


if __name__ == "__main__":
    main()
