"""Data fixtures taken from files the REFERENCE ships (data, not source; run once in the build container, where /root/reference is
mounted; the GPU box only sees the outputs):

  reference_shared_config.json   examples/reference.json merged with examples/shared/partial.json, i.e. what `make run_shared` feeds
                                 the reference (examples/Makefile:38-39 merges them with `gojq -s '.[0] * .[1]'`; gojq is not in
                                 this image, so the merge is done here: a flat dictionary update, there is no nested overlap)
  posterior_rows.npz             the 1000 posterior rows of examples/scratch/posterior.sqlite ("reusable as a
                                 realistic data fixture"): serial, particleIdx, posterior rank, 5 parameters, 7 metrics, column names

    python tests/golden/make_reference_fixtures.py
"""
import json
import os
import re
import sqlite3

import numpy as np

REF = "/root/reference/examples"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_json_with_comments(path):
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"(?m)//.*$", "", txt)
    return json.loads(txt)


def main():
    cfg = load_json_with_comments(os.path.join(REF, "reference.json"))
    cfg.update(load_json_with_comments(os.path.join(REF, "shared", "partial.json")))
    with open(os.path.join(HERE, "reference_shared_config.json"), "w") as f:
        json.dump(cfg, f, indent=2)
        f.write("\n")
    c = sqlite3.connect("file:%s?mode=ro" % os.path.join(REF, "scratch", "posterior.sqlite"), uri=True)
    pcols = [r[1] for r in c.execute("pragma table_info(parameters)")][2:]
    mcols = [r[1] for r in c.execute("pragma table_info(metrics)")][1:]
    rows = c.execute("select J.serial, J.particleIdx, J.posterior, J.smcSet, %s, %s from jobs J, parameters P, metrics M where "
                     "J.serial = P.serial and J.serial = M.serial order by J.particleIdx"
                     % (", ".join("P." + n for n in pcols), ", ".join("M." + n for n in mcols))).fetchall()
    a = np.array(rows, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "posterior_rows.npz"), serial=a[:, 0].astype(np.int64), particle_idx=a[:, 1].astype(np.int64),
                        posterior_rank=a[:, 2].astype(np.int64), smc_set=a[:, 3].astype(np.int64), parameters=a[:, 4:4 + len(pcols)],
                        metrics=a[:, 4 + len(pcols):], parameter_names=np.array(pcols), metric_names=np.array(mcols))
    print("wrote reference_shared_config.json (%d keys) and posterior_rows.npz (%d rows, %d parameters, %d metrics)"
          % (len(cfg), a.shape[0], len(pcols), len(mcols)))


if __name__ == "__main__":
    main()
