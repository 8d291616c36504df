"""
Persist a graph (prograph/utils/save.py:5-39 of the reference): pickle the DataFrame without
the cheap-to-recompute `Tokenized` column to `<directory><name>.pkl`.  Like the reference it
reports problems on stdout and returns True; a reloaded pickle that already carries a
`Neighbours` column skips graph construction (prograph/prograph.py:140-141).
"""


def save(pgraph, name=None, ext=".pkl", directory=None, ignored_cols=["Tokenized"]):
    file = "pgraph"
    if directory is None:
        if hasattr(pgraph, "file"):
            directory, file = pgraph.file.rsplit("/", 1)
            directory += "/"
        else:
            directory = "./"
    elif hasattr(pgraph, "file"):
        file = pgraph.file.rsplit("/", 1)[-1]
    if not name:
        name = file.rsplit(".", 1)[0] + "_pgraph"
    print(f"Saving Graph to {name + ext}")
    try:
        keep = [c for c in pgraph.graph if c not in ignored_cols]
        pgraph.graph[keep].to_pickle(directory + name + ext)
    except Exception as e:
        print("Error occurred during saving:", e)
    return True
