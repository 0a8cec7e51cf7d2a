"""Which build a summary under profiles/ was measured on: the fingerprint of the kernel sources and the commit."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neuralgaussiansplatting_amd import build_id  # noqa: E402


def source_record(raw):
    try:
        commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
        dirty = subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "neuralgaussiansplatting_amd/csrc"],
                                        text=True).strip()
        if dirty:
            commit += "+uncommitted kernel changes"
    except Exception:
        commit = "unrecorded"
    return {"kernel_sources_sha": build_id.kernel_sources_sha(), "commit": commit, "raw": raw}


def merged_table(path, source):
    """The summary file's current table, emptied if it belongs to another build."""
    import json
    table = {}
    if os.path.exists(path):
        with open(path) as f:
            table = json.load(f)
        if table.get("_source", {}).get("kernel_sources_sha") != source["kernel_sources_sha"]:
            table = {}
    table["_source"] = source
    return table
