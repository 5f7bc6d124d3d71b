"""Two bars per end-to-end parity check.

  1. the CONTRACT: BASELINE.json's 1e-3 abs on logits / probs / embeddings (fp32 modes), the drift bounds of the bf16 modes;
  2. the REGRESSION bar: FACTOR x the deviation this case measured on an MI355X when the floor was recorded
     (tests/golden/MANIFEST.json -> "measured_on_mi355x": {case: {output: max |delta|}}).  The kernels are deterministic
     (same arithmetic order on every box: tools/soak.py), so a case's deviation is a constant of the build; 10 x it (3 x for
     the bf16 drift) leaves room for a re-ordered accumulation and none for a lost mantissa bit: the contract bar alone
     would let a 100-fold numerical regression of fp32_split through (its floor is 1e-6 ... 1e-5).

A case that has no recorded floor yet fails loudly unless ACX_RECORD_FLOOR=<json path> is set; with it set every case
appends its measurements to that file (the contract bar still applies) -- `python tests/parity_floor.py merge <json>` then
folds the file into MANIFEST.json.  Run once per kernel change that is MEANT to move the numerics, and commit the result.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
MANIFEST = os.path.join(HERE, "golden", "MANIFEST.json")
KEY = "measured_on_mi355x"
ABS_MIN = 2e-6          # a floor of exactly 0 (bit-equal today) must not forbid the last-bit noise of a re-ordered sum


def _floors():
    with open(MANIFEST) as f:
        return json.load(f).get(KEY, {})


def check(case, res, contract, factor=10.0):
    """res: {output: max abs deviation}; contract: one bar or {output: bar}.  Asserts both bars (or records, see above)."""
    bars = contract if isinstance(contract, dict) else {k: contract for k in res}
    for k, v in res.items():
        assert v < bars[k], "%s: %s deviates %.3e, contract bar %.1e" % (case, k, v, bars[k])
    rec = os.environ.get("ACX_RECORD_FLOOR")
    if rec:
        data = {}
        if os.path.exists(rec):
            with open(rec) as f:
                data = json.load(f)
        data[case] = {k: float(v) for k, v in res.items()}
        with open(rec, "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)
        return
    floor = _floors().get(case)
    assert floor is not None, "no recorded floor for case %r: run the GPU suite once with ACX_RECORD_FLOOR=<file>, then " \
                              "`python tests/parity_floor.py merge <file>`" % case
    for k, v in res.items():
        bar = max(factor * floor[k], ABS_MIN)
        assert v <= bar, "%s: %s deviates %.3e = %.1f x the recorded floor %.3e (regression bar %.1f x)" % (
            case, k, v, v / max(floor[k], 1e-30), floor[k], factor)


def merge(path):
    with open(path) as f:
        new = json.load(f)
    with open(MANIFEST) as f:
        m = json.load(f)
    m.setdefault(KEY, {}).update(new)
    with open(MANIFEST, "w") as f:
        json.dump(m, f, indent=1, sort_keys=True)
    print("merged %d cases into %s" % (len(new), MANIFEST))


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "merge":
        merge(sys.argv[2])
    else:
        sys.exit(__doc__)
