#!/usr/bin/env python3
"""Build tests/golden/*_csr.npz from the topology DATA files the reference holds.

Runs only in the build container (reads /root/reference as data, executes no
reference code).  Pattern construction restates utils.py:49-52: symmetrise the
edge list and add the identity; only the pattern matters (layers.py:41,129), so
the D^-1/2 (A+I) D^-1/2 values of utils.py:73-79 are not kept.

  cora      data/cora/cora.cites  (utils.py:27-29).  cora.content is a missing
            blob, so the reference's node order (row order of cora.content,
            utils.py:25-26) is unknown: ids are ranked in ascending order here.
  citeseer  citeseer_dgl/adj_sparse.npz (utils.py:45), COO, self loops included
  pubmed    pubmed_dgl/adj_sparse.npz   (utils.py:45)
  ppi       data/ppi/*_graph_id.npy -> per-graph node counts; data/ppi/valid_feats.npy -> a 1612-row sample
  labels    {citeseer,pubmed}_dgl/labels.pt, idx_{train,val,test}.pt (utils.py:40-43) -> *_labels.npz

Output: rowptr int32 [N+1], col int32 [E] (sorted within a row).
"""
import os
import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def to_csr(r, c, n):
    rr = np.concatenate([r, c, np.arange(n)]).astype(np.int64)
    cc = np.concatenate([c, r, np.arange(n)]).astype(np.int64)
    key = np.unique(rr * n + cc)
    rr = key // n
    cc = (key % n).astype(np.int32)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rowptr, rr + 1, 1)
    return np.cumsum(rowptr).astype(np.int32), cc


def main():
    e = np.loadtxt(f"{REF}/data/cora/cora.cites", dtype=np.int64)
    ids = np.unique(e)
    rank = {v: i for i, v in enumerate(ids)}
    r = np.array([rank[v] for v in e[:, 0]]); c = np.array([rank[v] for v in e[:, 1]])
    sets = {"cora": to_csr(r, c, len(ids))}
    for name in ("citeseer", "pubmed"):
        z = np.load(f"{REF}/{name}_dgl/adj_sparse.npz", allow_pickle=False)
        n = int(z["shape"][0])
        sets[name] = to_csr(z["row"].astype(np.int64), z["col"].astype(np.int64), n)
    for name, (rowptr, col) in sets.items():
        deg = np.diff(rowptr)
        print(f"{name}: N={len(rowptr)-1} E={len(col)} deg min/med/mean/max = "
              f"{deg.min()}/{int(np.median(deg))}/{deg.mean():.2f}/{deg.max()}")
        np.savez_compressed(os.path.join(OUT, f"{name}_csr.npz"), rowptr=rowptr, col=col)
    # labels and splits of the DGL dumps (utils.py:40-43); weights_only=True executes nothing from the files
    import torch
    for name in ("citeseer", "pubmed"):
        d = f"{REF}/{name}_dgl"
        ld = lambda f: torch.load(f"{d}/{f}.pt", weights_only=True).numpy()  # noqa: E731
        np.savez_compressed(os.path.join(OUT, f"{name}_labels.npz"), labels=ld("labels").astype(np.int16),
                            idx_train=ld("idx_train").astype(np.int32), idx_val=ld("idx_val").astype(np.int32),
                            idx_test=ld("idx_test").astype(np.int32))
    # PPI: per-graph node counts are all that survives offline (graph JSONs are missing blobs)
    counts = {}
    for split in ("train", "valid", "test"):
        gid = np.load(f"{REF}/data/ppi/{split}_graph_id.npy", allow_pickle=False)
        _, cnt = np.unique(gid, return_counts=True)
        counts[split] = cnt.astype(np.int32)
        print(f"ppi {split}: {len(cnt)} graphs, nodes {cnt.min()}..{cnt.max()}, total {cnt.sum()}")
    np.savez_compressed(os.path.join(OUT, "ppi_graph_sizes.npz"), **counts)
    # PPI node features: the first 591 + 1021 rows of valid_feats.npy (load_data_ppi.py:110-121 reads these files;
    # train_feats.npy is a missing blob), enough for one PPI-shaped batch of two graphs
    vf = np.load(f"{REF}/data/ppi/valid_feats.npy", allow_pickle=False)
    np.savez_compressed(os.path.join(OUT, "ppi_feats_sample.npz"), feats=vf[:591 + 1021].astype(np.float32))
    print("ppi feats sample:", vf[:1612].shape, "range", float(vf[:1612].min()), float(vf[:1612].max()))


if __name__ == "__main__":
    main()
