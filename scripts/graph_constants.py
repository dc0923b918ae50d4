#!/usr/bin/env python3
"""Extracts the numeric constants of the legacy AudioMPS training graph that the reference ships as
/root/reference/logging/graph.pbtxt (the serialized graph of the notebook's `sine_model`, D=5, B=8, T=4096) into
tests/golden/legacy_graph_constants.json.

These are the only reference-held NUMBERS on the hot path: the loop bound, the factor 2 of the
expectation, the exponent and the divisor of the squared-error loss, the -1j of the Hamiltonian term, the divisor of the
R^T R term, delta_t in both places it enters the update, the floor of the normalisation, and Adam's hyper-parameters.
tests/test_oracle.py::test_legacy_oracle_uses_the_graphs_constants evaluates one scan step from these numbers alone
and compares it with oracle/cmps_oracle.py::legacy_loss_and_grads.

Runs only where /root/reference exists (the build container); the JSON it writes is data and travels with the repo.
    python scripts/graph_constants.py [/root/reference/logging/graph.pbtxt]
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/logging/graph.pbtxt"

# graph node name -> (key in the JSON, what it is)
WANTED = {
    "sine_model/loss_fold/while/Less/y": ("loop_bound", "tf.foldl trip count = T - 1"),
    "sine_model/loss_fold/while/expectation/mul/x": ("expectation_factor", "e = FACTOR * Re<psi|R|psi>"),
    "sine_model/loss_fold/while/pow/y": ("loss_exponent", "(x - e) ** EXPONENT"),
    "sine_model/loss_fold/while/truediv/y": ("loss_divisor", "... / DIVISOR"),
    "sine_model/loss_fold/while/update_ancilla/mul/x": ("hamiltonian_factor", "FACTOR * H  (complex64)"),
    "sine_model/loss_fold/while/update_ancilla/truediv/y": ("dissipator_divisor", "R^T R / DIVISOR"),
    "sine_model/loss_fold/while/update_ancilla/mul_1/x": ("delta_t_Q", "Q = DT * (-i H - R^T R / 2)"),
    "sine_model/loss_fold/while/update_ancilla/mul_2/x": ("delta_t_signal", "DT * x * (R psi)"),
    "sine_model/loss_fold/while/update_ancilla/normalize/Maximum/y": ("norm_floor", "max(sum |psi|^2, FLOOR)"),
    "Adam/learning_rate": ("adam_learning_rate", "tf.train.AdamOptimizer"),
    "Adam/beta1": ("adam_beta1", ""),
    "Adam/beta2": ("adam_beta2", ""),
    "Adam/epsilon": ("adam_epsilon", ""),
}


def const_nodes(text):
    """Yields (name, dtype, [values], first line number) for every scalar Const node."""
    pos = 0
    for m in re.finditer(r"^node \{\n  name: \"([^\"]+)\"\n  op: \"Const\"\n", text, flags=re.M):
        end = text.find("\nnode {", m.end())
        body = text[m.start():end if end != -1 else len(text)]
        dt = re.search(r"dtype: (DT_\w+)", body)
        vals = re.findall(r"\b(?:float_val|int_val|scomplex_val|double_val|int64_val): ([-+.\deE]+|-?inf|nan)", body)
        line = text.count("\n", 0, m.start()) + 1
        yield m.group(1), dt.group(1) if dt else None, [float(v) for v in vals], line


def main():
    with open(SRC) as fh:
        text = fh.read()
    out = {"source": "logging/graph.pbtxt of AustenLamacraft/audio-mps (legacy AudioMPS training graph, notebook sine_model)",
           "constants": {}}
    for name, dtype, vals, line in const_nodes(text):
        if name in WANTED:
            key, what = WANTED[name]
            v = vals
            if dtype == "DT_COMPLEX64":
                v = {"re": vals[0], "im": vals[1]}
            elif len(vals) == 1:
                v = int(vals[0]) if dtype == "DT_INT32" else vals[0]
            out["constants"][key] = {"value": v, "dtype": dtype, "node": name, "line": line, "meaning": what}
    missing = [k for k, _ in WANTED.values() if k not in out["constants"]]
    if missing:
        raise SystemExit(f"not found in {SRC}: {missing}")
    dst = os.path.join(ROOT, "tests", "golden", "legacy_graph_constants.json")
    with open(dst, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print(f"wrote {dst}")
    for k, v in sorted(out["constants"].items()):
        print(f"  {k:22s} {v['value']!r:28}  line {v['line']}")


if __name__ == "__main__":
    main()
