#!/usr/bin/env python3
"""Golden vectors for the host half of the self-training driver and the evaluation metrics:
  * the reference's OWN `generate_self_train_data` (AST-extracted from drl_classifier_ec_mmd_final_mul.py:734-799 at generation
    time, one-line `DataFrame.append` shim for pandas >= 2) on synthetic documents with a stand-in model whose get_pair_preds
    returns fixed values -- the rounded 0. / 1. lists the reference's own model returns (:282) and fractional scores -- for the
    three strategies, `random.seed` fixed before every call (the "random" strategy draws `randint` once per pair after the first);
  * sklearn's precision / recall / F1 (binary; ref :868-870) on a few label / prediction vectors incl. the zero-division cases.
Inputs and outputs only are stored (tests/golden/selftrain.json).   python tests/golden/gen_golden_selftrain.py"""
import ast, json, os, random, warnings
import pandas as pd
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def reference_fn():
    src = open(os.path.join(REF, "drl_classifier_ec_mmd_final_mul.py"), encoding="utf8").read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "generate_self_train_data"]
    ns = dict(pd=pd, torch=torch, randint=random.randint, device="cpu")
    exec(compile(ast.Module(body=fn, type_ignores=[]), "<reference:generate_self_train_data>", "exec"), ns)
    return ns["generate_self_train_data"]


class Model:
    def __init__(self, scores): self.scores = scores
    def eval(self): pass
    def get_pair_preds(self, ids, att, tt): return [[float(s)] for s in self.scores]


def case(name, sizes, scores, strategy, seed):
    n = sum(sizes)
    assert n == len(scores)
    pairs = ["doc-pair-%03d" % i for i in range(n)]
    emotions = [(7 * i + 3) % 6 for i in range(n)]
    df = pd.DataFrame({"pair": pairs, "label": [0] * n, "emotion": emotions})
    z = torch.zeros((n, 4), dtype=torch.long)
    loader = [{"input_ids": z, "attention_masks": z + 1, "token_type_ids": z}]
    random.seed(seed)
    out = reference_fn()(list(sizes), df, loader, Model(scores), strategy)
    rows = [[r["pair"], int(r["label"]), None if r["emotion"] is None or pd.isna(r["emotion"]) else int(r["emotion"])] for _, r in out.iterrows()]
    return dict(name=name, sizes=list(sizes), scores=[float(s) for s in scores], strategy=strategy, seed=seed, pairs=pairs, emotions=emotions,
                columns=list(out.columns), rows=rows)


if __name__ == "__main__":
    if not hasattr(pd.DataFrame, "append"):
        pd.DataFrame.append = lambda self, row, ignore_index=True: pd.concat([self, pd.DataFrame([row])], ignore_index=True)
    rs = random.Random(5)
    sizes = [3, 1, 5, 2, 6, 1, 4]
    n = sum(sizes)
    rounded = [float(rs.random() < 0.4) for _ in range(n)]              # what the reference's own get_pair_preds returns (:282)
    frac = [round(rs.random(), 3) for _ in range(n)]
    allzero = [0.0] * n
    cases = []
    for strategy in ("random", "extreme", "threshold"):
        cases.append(case(strategy + "_rounded", sizes, rounded, strategy, 42))
        cases.append(case(strategy + "_fractional", sizes, frac, strategy, 7))
        cases.append(case(strategy + "_all_zero", sizes, allzero, strategy, 1))
    from sklearn.metrics import precision_score, recall_score, f1_score
    metrics = []
    vecs = [([1, 0, 1, 1, 0, 0, 1], [1, 0, 0, 1, 1, 0, 1]), ([0, 0, 0], [0, 0, 0]), ([1, 1, 0], [0, 0, 0]), ([0, 0, 1], [1, 1, 0]),
            ([1] * 5 + [0] * 9, [1, 0, 1, 1, 0] + [0, 1, 0, 0, 0, 0, 1, 0, 0]), ([1, 1, 1], [1, 1, 1])]
    for y, p in vecs:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            yy, pp = [[v] for v in y], [[float(v)] for v in p]          # the nested lists train() passes (:860-865)
            metrics.append(dict(labels=y, preds=p, precision=float(precision_score(yy, pp, average="binary")),
                                recall=float(recall_score(yy, pp, average="binary")), f1=float(f1_score(yy, pp, average="binary"))))
    import sklearn
    json.dump(dict(meta=dict(pandas=pd.__version__, sklearn=sklearn.__version__, source="drl_classifier_ec_mmd_final_mul.py:734-799, :868-870"),
                   self_train=cases, metrics=metrics), open(os.path.join(HERE, "selftrain.json"), "w"), indent=1)
    print({c["name"]: len(c["rows"]) for c in cases})
