"""Hash of the sources the hot kernels are built from.  profiles/pmc_traffic.json (the recorded PMC counters bench.py
replays as roofline.traffic / roofline.compute) carries the hash of the tree it was measured on; bench.py reports the
counters only while the working tree still has that hash -- a kernel change that forgets to refresh the profile gets
`traffic: null` with the reason instead of stale numbers."""
import hashlib
import os

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
HOT_SOURCES = ("trx_kernel_nb.hip", "trx_nb_asm.inc", "trx_kernel4.hip", "trx_k4_common.h", "trx_device.h", "trx_tables.h")


def hot_kernel_source_hash(csrc=CSRC):
    h = hashlib.sha256()
    for name in HOT_SOURCES:
        with open(os.path.join(csrc, name), "rb") as f:
            data = f.read()
        h.update(name.encode() + b"\0" + hashlib.sha256(data).digest())
    return h.hexdigest()[:16]


def replay_counters(pmc_path, csrc=CSRC):
    """(record, reason): the recorded counters if they belong to the current sources, else (None, why)."""
    import json
    if not os.path.exists(pmc_path):
        return None, "no recorded profile (profiles/pmc_traffic.json)"
    try:
        j = json.load(open(pmc_path))
    except Exception as e:                                  # noqa: BLE001
        return None, f"unreadable profile: {e}"
    want = hot_kernel_source_hash(csrc)
    if j.get("source_hash") != want:
        return None, (f"recorded profile {j.get('tag', '?')} was measured on kernel sources {j.get('source_hash', 'without a hash')}, "
                      f"the working tree has {want}: re-run tools/run_profiles.sh + profiles/summarize.py")
    return j, None
