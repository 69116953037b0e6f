"""Extract the literal input/expected arrays of the reference's test_rollout_postprocess
(test/testJAXTrainer.py:91-389) as DATA: the source is read as text, the `jnp.array([...])` literals are
parsed with ast.literal_eval (nothing of the reference is imported or executed) and written to
tests/golden/rollout_postprocess.json together with the (role, use_unified_tree) of each assertion.
The trainer config of that test (test/jax_test_config.yml / hironaka/jax/jax_config.yml) has dimension 3
and discount 0.99."""
import ast
import json
import os
import re

REF = os.environ.get("HIRONAKA_REFERENCE", "/root/reference")
src = open(os.path.join(REF, "test", "testJAXTrainer.py")).read()
i = src.index("def test_rollout_postprocess")
j = src.index("    def test_", i + 10)
body = src[i:j]


def arrays_in(text):
    """all top-level jnp.array(<literal>) literals in order, as nested lists"""
    out, pos = [], 0
    while True:
        k = text.find("jnp.array(", pos)
        if k < 0:
            return out
        depth, e = 0, k + len("jnp.array")
        for e in range(k + len("jnp.array"), len(text)):
            if text[e] == "(":
                depth += 1
            elif text[e] == ")":
                depth -= 1
                if depth == 0:
                    break
        lit = text[k + len("jnp.array("):e]
        lit = re.sub(r",\s*dtype=[\w.]+", "", lit)
        out.append(ast.literal_eval(lit.strip()))
        pos = e + 1


arrs = arrays_in(body)
asserts = re.findall(r'assert jnp\.all\(jnp\.isclose\(self\.trainer\.rollout_postprocess\(rollout, "(\w+)", use_unified_tree=(\w+)\)', body)
# order of appearance in the test: rollout#1 (obs, policy, value), v1; rollout#2 (obs, policy, value), v2;
# [rollout#2 sliced [:, 1:]], v3; rollout#3 (obs, policy, value), v4; obs#4 (policy/value reused), v5
assert len(arrs) == 15 and len(asserts) == 5, (len(arrs), len(asserts))
obs1, _, _, v1, obs2, _, _, v2, v3, obs3, _, _, v4, obs4, v5 = arrs
if len(obs1) == 1 and isinstance(obs1[0][0][0], list):  # rollout#1's literal carries a leading device axis of 1
    obs1 = obs1[0]
obs = [obs1, obs2, [g[1:] for g in obs2], obs3, obs4]
exp = [v1, v2, v3, v4, v5]
cases = [{"obs": o, "role": a[0], "unified": a[1] == "True", "expected": e} for o, a, e in zip(obs, asserts, exp)]
out = {"source": "test/testJAXTrainer.py:91-389 test_rollout_postprocess (isclose)", "dimension": 3, "discount": 0.99,
       "cases": cases}
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rollout_postprocess.json"), "w") as f:
    json.dump(out, f, separators=(",", ":"))
print(len(cases), "cases;", [(c["role"], c["unified"], len(c["obs"]), len(c["obs"][0])) for c in cases])
