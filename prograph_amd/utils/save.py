"""
Persist a graph to `<directory><name>.pkl` (the reference's prograph/utils/save.py:5-39): the
DataFrame is pickled without the cheap-to-recompute `Tokenized` column.  A reloaded pickle that
already carries a `Neighbours` column skips graph construction (prograph/prograph.py:140-141).
Like the reference, problems are reported on stdout and the function still returns True.

`graphs="csr"` (not in the reference) keeps the N per-row tuples out of the pickle: every graph column
that has a device-resident CSR / kNN form (`Prograph.csr_graphs`) is written as flat arrays to the
side-car `<name>.graphs.npz` instead and dropped from the pickled frame; `Prograph("<name>.pkl")`
finds the side-car, restores the device graphs and the columns from it and, as with the reference's
own pickles, does not build the `Neighbours` graph again.
"""
import os


def save(pgraph, name=None, ext=".pkl", directory=None, ignored_cols=["Tokenized"], graphs="tuples"):
    source = getattr(pgraph, "file", None)
    stem = os.path.splitext(os.path.basename(source))[0] if source else "pgraph"
    if directory is None:
        directory = (os.path.dirname(source) + "/") if source and os.path.dirname(source) else "./"
    target = directory + (name or stem + "_pgraph") + ext
    print(f"Saving Graph to {os.path.basename(target)}")
    try:
        ignored = list(ignored_cols)
        sidecar = os.path.splitext(target)[0] + ".graphs.npz"
        if graphs == "csr":
            from ..graph import save_graphs, fingerprint
            live = {n: g for n in pgraph.csr_graphs if n in pgraph.graph and (g := pgraph._device_graph_any(n)) is not None}
            save_graphs(sidecar, live, tokens_fingerprint=fingerprint(pgraph.tokenized))
            ignored += list(live)
        elif os.path.exists(sidecar):
            os.remove(sidecar)          # a side-car of an earlier graphs="csr" save would shadow this pickle's columns
        columns = [c for c in pgraph.graph.columns if c not in ignored]
        pgraph.graph[columns].to_pickle(target)
    except Exception as err:
        print("Error occurred during saving:", err)
    return True
