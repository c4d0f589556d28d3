#!/usr/bin/env python3
"""Golden vectors for the host data path: runs the reference's OWN `read_ECPE_data` (AST-extracted from
drl_classifier_ec_mmd_final_mul.py:631-731 at generation time, with a one-line `DataFrame.append` shim for
pandas >= 2) on the small ECPE-format samples committed under tests/golden/ecpe/ (written for this repo, not
copied from the reference's data) and, for an optional cross-check, on the reference's real files (row counts
and a digest only -- no reference data is copied).   python tests/golden/gen_golden_data.py"""
import ast, hashlib, json, os, random, re, sys, types
import pandas as pd

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def reference_reader(language):
    src = open(os.path.join(REF, "drl_classifier_ec_mmd_final_mul.py"), encoding="utf8").read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "read_ECPE_data"]
    ns = dict(pd=pd, re=re, random=random, opt=types.SimpleNamespace(language=language))
    exec(compile(ast.Module(body=fn, type_ignores=[]), "<reference:read_ECPE_data>", "exec"), ns)
    return ns["read_ECPE_data"]


def reference_reader_en_script():
    """read_ECPE_data of drl_classifier_en.py (:748-813): two return values, no emotion column."""
    src = open(os.path.join(REF, "drl_classifier_en.py"), encoding="utf8").read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "read_ECPE_data"]
    ns = dict(pd=pd, re=re, random=random)
    exec(compile(ast.Module(body=fn, type_ignores=[]), "<reference:drl_classifier_en.read_ECPE_data>", "exec"), ns)
    return ns["read_ECPE_data"]


def run_en_script(path, test):
    random.seed(42)
    df, sizes = reference_reader_en_script()(path, test=test)
    h = hashlib.sha1()
    for p, l in zip(df["pair"], df["label"]):
        h.update(("%s|%d\n" % (p, int(l))).encode("utf8"))
    return dict(rows=len(df), docs_pair_size=[int(s) for s in sizes], digest=h.hexdigest(), columns=list(df.columns),
                head=[[str(p), int(l)] for p, l in zip(df["pair"][:6], df["label"][:6])])


def digest(df):
    h = hashlib.sha1()
    for p, l, e in zip(df["pair"], df["label"], df["emotion"]):
        h.update(("%s|%d|%d\n" % (p, int(l), int(e))).encode("utf8"))
    return h.hexdigest()


def run(path, language, test):
    random.seed(42)
    df, sizes, unpred = reference_reader(language)(path, test=test)
    return dict(rows=len(df), docs_pair_size=[int(s) for s in sizes], num_unpred=int(unpred), digest=digest(df),
                head=[[str(p), int(l), int(e)] for p, l, e in zip(df["pair"][:6], df["label"][:6], df["emotion"][:6])])


if __name__ == "__main__":
    if not hasattr(pd.DataFrame, "append"):
        pd.DataFrame.append = lambda self, row, ignore_index=True: pd.concat([self, pd.DataFrame([row])], ignore_index=True)
    out = {"samples": {}, "reference_files": {}}
    for name, lang, test in (("sample_zh_train.txt", "zh", False), ("sample_zh_test.txt", "zh", True), ("sample_en_train.txt", "en", False)):
        out["samples"][name] = dict(language=lang, test=test, **run(os.path.join(HERE, "ecpe", name), lang, test))
    out["samples_en_script"] = {"sample_en_train.txt:train": dict(test=False, **run_en_script(os.path.join(HERE, "ecpe", "sample_en_train.txt"), False)),
                                "sample_en_train.txt:test": dict(test=True, **run_en_script(os.path.join(HERE, "ecpe", "sample_en_train.txt"), True))}
    for rel, lang, test in (("domains/THUCTC_multiple/society_num.txt", "zh", False), ("pair_data/emotion/education.txt", "zh", True)):
        r = run(os.path.join(REF, rel), lang, test)
        r.pop("head")
        r["docs"] = len(r.pop("docs_pair_size"))
        out["reference_files"][rel] = dict(language=lang, test=test, **r)
    json.dump(out, open(os.path.join(HERE, "ecpe_data.json"), "w"), ensure_ascii=False, indent=1)
    print(json.dumps({k: {n: (v["rows"], v.get("num_unpred")) for n, v in d.items()} for k, d in out.items()}))
