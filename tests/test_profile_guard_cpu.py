"""The recorded PMC counters bench.py replays (profiles/pmc_traffic.json) are tied to the sources of the hot kernels they were
measured on (osmo_trx_amd/srchash.py): replayed while the hash matches, `None` + the reason otherwise."""
import json
import os
import shutil

from osmo_trx_amd import srchash

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _copy_sources(dst):
    os.makedirs(dst)
    for name in srchash.HOT_SOURCES:
        shutil.copyfile(os.path.join(srchash.CSRC, name), os.path.join(dst, name))


def test_counters_replayed_only_on_the_sources_they_were_measured_on(tmp_path):
    csrc = str(tmp_path / "csrc")
    _copy_sources(csrc)
    h = srchash.hot_kernel_source_hash(csrc)
    assert h == srchash.hot_kernel_source_hash()                       # a copy of the tree hashes like the tree
    pmc = tmp_path / "pmc_traffic.json"
    pmc.write_text(json.dumps({"tag": "t", "source_hash": h, "hbm_bytes_per_burst": 3150.0}))
    rec, why = srchash.replay_counters(str(pmc), csrc)
    assert why is None and rec["hbm_bytes_per_burst"] == 3150.0
    # one byte of one hot source changes: the profile is stale
    with open(os.path.join(csrc, "trx_kernel_nb.hip"), "a") as f:
        f.write("\n")
    rec, why = srchash.replay_counters(str(pmc), csrc)
    assert rec is None and "re-run tools/run_profiles.sh" in why and h in why
    # a profile from before the hash existed, a missing one, an unreadable one
    pmc.write_text(json.dumps({"tag": "r05d", "hbm_bytes_per_burst": 3159.0}))
    assert srchash.replay_counters(str(pmc), csrc)[0] is None
    assert srchash.replay_counters(str(tmp_path / "none.json"), csrc)[0] is None
    pmc.write_text("{")
    assert srchash.replay_counters(str(pmc), csrc)[0] is None


def test_the_committed_profile_is_either_current_or_reported_stale():
    rec, why = srchash.replay_counters(os.path.join(ROOT, "profiles", "pmc_traffic.json"))
    assert (rec is None) != (why is None)
    if rec is not None:
        assert rec["source_hash"] == srchash.hot_kernel_source_hash() and rec["hbm_bytes_per_burst"] > 3000
