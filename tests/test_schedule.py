"""CPU checks of the render forward's schedule table (vampire_amd.ops.render_forward_plan): every combination
of its five inputs gives a well-formed list of calls."""
import itertools

from vampire_amd import _capi
from vampire_amd.ops import render_forward_plan


def test_merged_schedules():
    """The one-launch render forward stands for "cam" + "bev": exactly where the one-kernel camera forward with early
    termination runs forward-only, never otherwise."""
    for train, two, prep_ok, direct, ert in itertools.product((False, True), repeat=5):
        plan = render_forward_plan(train, two, prep_ok, direct, ert, merged=True)
        ops = [p[0] for p in plan]
        if direct and ert and not train:
            assert ops == ["render"] and plan[0][1] == "cur" and plan[0][2] == 0, ((train, two, prep_ok, direct, ert), ops)
        elif direct and ert and train and prep_ok:
            # training: the one launch also draws the backward's cell ranks; scan + heavy list behind it
            assert ops == ["render", "prep"] and plan[0][1] == "cur" and plan[0][2] == _capi.VAMP_RENDERFWD_RANK
            assert plan[1][2] == _capi.VAMP_CAMPREP_RANKED and plan[1][1] == "cur"
        else:
            assert plan == render_forward_plan(train, two, prep_ok, direct, ert), (train, two, prep_ok, direct, ert)


def test_every_schedule_is_well_formed():
    for train, two, prep_ok, direct, ert in itertools.product((False, True), repeat=5):
        plan = render_forward_plan(train, two, prep_ok, direct, ert)
        ops = [p[0] for p in plan]
        key = (train, two, prep_ok, direct, ert)
        assert ops.count("bev") == 1 and ops.count("cam") == 1, (key, ops)
        recorded = set()
        for op, where, flags, waits, records in plan:
            assert where in ("cur", "side") and (two or where == "cur"), (key, op, where)
            assert set(waits) <= recorded, (key, op, waits)          # an event is recorded before it is waited for
            recorded |= set(records)
        cam = next(p for p in plan if p[0] == "cam")
        assert cam[1] == "cur"                                       # the outputs appear on the caller's stream
        assert bool(cam[2] & _capi.VAMP_CAMFWD_DIRECT) == direct, key
        # a termination pre-pass exactly when the planned march runs with early termination, and then the march
        # (and the prepare pass) are told the table is there and wait for it
        assert ("term" in ops) == (ert and not direct), key
        assert bool(cam[2] & _capi.VAMP_CAMFWD_TERM_VALID) == ("term" in ops), key
        if "pack" in ops:
            assert cam[2] & _capi.VAMP_CAMFWD_PACKED_VALID and "packed" in cam[3], key
            assert ops.index("term") < ops.index("pack") < ops.index("cam"), key
        # the prepare pass only in front of the cell-list backward of a two-stream training step
        assert ("prep" in ops) == (train and two and prep_ok), key
        if "prep" in ops:
            prep = next(p for p in plan if p[0] == "prep")
            assert prep[1] == "side"
            has_table = direct or ert
            assert bool(prep[2] & _capi.VAMP_CAMPREP_TERM_VALID) == has_table, key
            if has_table:       # the table's producer is ahead of it: on its own stream, or through an event
                src = "cam" if direct else "term"
                producer = next(p for p in plan if p[0] == src)
                assert ops.index(src) < ops.index("prep") and (producer[1] == "side" or set(prep[3]) & set(producer[4])), key
