"""
Persist a graph to `<directory><name>.pkl` (the reference's prograph/utils/save.py:5-39): the
DataFrame is pickled without the cheap-to-recompute `Tokenized` column.  A reloaded pickle that
already carries a `Neighbours` column skips graph construction (prograph/prograph.py:140-141).
Like the reference, problems are reported on stdout and the function still returns True.
"""
import os


def save(pgraph, name=None, ext=".pkl", directory=None, ignored_cols=["Tokenized"]):
    source = getattr(pgraph, "file", None)
    stem = os.path.splitext(os.path.basename(source))[0] if source else "pgraph"
    if directory is None:
        directory = (os.path.dirname(source) + "/") if source and os.path.dirname(source) else "./"
    target = directory + (name or stem + "_pgraph") + ext
    print(f"Saving Graph to {os.path.basename(target)}")
    try:
        columns = [c for c in pgraph.graph.columns if c not in ignored_cols]
        pgraph.graph[columns].to_pickle(target)
    except Exception as err:
        print("Error occurred during saving:", err)
    return True
