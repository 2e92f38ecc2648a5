"""CPU: A/B switches of the specialised kernel (mmpc_fast.h) in the host lane-emulation build - the rarely or never taken
branches and the alternative code paths behind the shipped ones must give the shipped results."""
import numpy as np

from oracle import nlp, synth
import emu_helper


def _run(defs=(), B=16, **kw):
    d = synth.make_batch(B)
    return emu_helper.solve_batch(nlp.WholeBodyParams(), d["x_init"], d["traj_ref"], d["u_ref"], np.zeros((B, 20, 5)), d["obs"],
                                  fast=True, defs=defs, **kw)


def test_lazy_safeguard_slow_path_is_the_eager_clamp():
    """The multiplier safeguard of the first trial is a range check per row; the exact clamp runs in a second phase only when a
    row needs it (never on the bench batches).  Forced after every first trial it must leave every bit as it is, and both must be
    what the eager clamp of rounds 1-3 gives."""
    ref = _run()
    forced = _run(("MMPC_SAFEGUARD_FORCE=1",))
    eager = _run(("MMPC_SAFEGUARD_LAZY=0",))
    assert (ref["status"] == 0).all()
    for o in (forced, eager):
        assert np.array_equal(ref["iters"], o["iters"])
        assert np.array_equal(ref["X"], o["X"]) and np.array_equal(ref["U"], o["U"]) and np.array_equal(ref["s"], o["s"])


def test_alternative_code_paths_agree():
    """The plain pair map; the sequential second pivot of a pair leg together with the two reciprocals per row of D2: the code these
    switches replaced - same iteration counts, same solutions to rounding (sums and products are formed in another order)."""
    ref = _run()
    for defs in (("MMPC_PADMAP=0",), ("MMPC_LEG_DET=0", "MMPC_D2_ONE_RCP=0")):
        o = _run(defs)
        assert (o["status"] == 0).all(), defs
        assert (np.abs(ref["iters"] - o["iters"]) <= 1).all(), defs
        assert np.abs(ref["X"] - o["X"]).max() < 1e-9 and np.abs(ref["U"] - o["U"]).max() < 1e-8, defs
