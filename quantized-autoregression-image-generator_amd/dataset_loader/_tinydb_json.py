"""Reader for the TinyDB JSON files the reference's tools write
({"_default": {"1": {...}, "2": {...}}}; generate_fmap_dataset.py:60-72, README.md:80),
without the tinydb package (absent here)."""
import json


def read_all(path, table="_default"):
    with open(path, "r") as f:
        data = json.load(f)
    rows = data.get(table, {})
    return [rows[k] for k in sorted(rows, key=lambda s: int(s))]


def write_all(path, records, table="_default"):
    with open(path, "w") as f:
        json.dump({table: {str(i + 1): r for i, r in enumerate(records)}}, f)
