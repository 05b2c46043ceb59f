"""CPU checks of the scenario tables (flightbatch/scenario.py): the blob's layout against include/flightbatch.h, the builder's refusals, and the
host interpreter's semantics (one transition per evaluation, `always` actions first, values read when their action runs)."""
import numpy as np
import pytest


def _arrays(fb, n):
    K = fb.K
    return dict(phase=np.zeros(n, np.int64), since=np.zeros(n, np.int64), step=0, cu=np.zeros((K["FB_NCU"], n)), cs=np.zeros((K["FB_NCS"], n)),
                u=np.zeros((K["FB_NU"], n)), ui=np.zeros(n, np.int32), s=np.zeros((K["FB_NS"], n), np.int32), active=np.ones(n, bool),
                h_e=np.zeros(n), psi=np.zeros(n), theta=np.zeros(n), phi=np.zeros(n), chi=np.zeros(n), EAS=np.zeros(n), clm=np.zeros(n), on_gnd=np.zeros(n))


def test_blob_layout_and_builder_refusals(fb):
    from flightbatch import scenario as sc
    K = fb.K
    scn = sc.Scenario(n_par=3, n_rec=2)
    A, B = scn.phase("a"), scn.phase("b")
    scn.always(A, [sc.u("FLAPS", 1.0)])
    scn.when(A, sc.src.H_E - sc.par(2) < 6.0, [sc.cu("BETA_REF", sc.wrap_to_pi(sc.src.PSI - sc.cs_("SEG_CHI_REF") + sc.cs_("SEG_DCHI"))), sc.rec(1, sc.src.T)], then=B)
    b = scn.pack()
    assert b[0] == 5000001.0 and list(b[1:6]) == [2, 1, 3, 3, 2]
    assert b.size == K["FB_SCN_HDR"] + 2 * K["FB_SCN_PHASE_REC"] + 1 * K["FB_SCN_RULE_REC"] + 3 * K["FB_SCN_ACT_REC"]
    ph = b[K["FB_SCN_HDR"]:K["FB_SCN_HDR"] + 8].reshape(2, 4)
    assert list(ph[0]) == [0, 1, 0, 1] and list(ph[1]) == [3, 0, 1, 0]
    ru = b[K["FB_SCN_HDR"] + 8:K["FB_SCN_HDR"] + 16]
    assert list(ru) == [K["FB_SCN_SRC_H_E"], 0, K["FB_SCN_LT"], 6.0, 2, 1, 2, B]
    ac = b[K["FB_SCN_HDR"] + 16:].reshape(3, K["FB_SCN_ACT_REC"])
    assert ac[1][0] == K["FB_SCN_DST_CU"] and ac[1][1] == K["FB_CU_BETA_REF"] and ac[1][2] == 1 and ac[1][4] == 3
    assert list(ac[1][5:14]) == [K["FB_SCN_SRC_PSI"], 0, 1.0, K["FB_SCN_SRC_CS"], K["FB_CS_SEG_CHI_REF"], -1.0, K["FB_SCN_SRC_CS"], K["FB_CS_SEG_DCHI"], 1.0]
    with pytest.raises(ValueError):
        sc.src.H_E + sc.src.PSI < 1.0          # two plain sources in a condition
    with pytest.raises(ValueError):
        sc.src.H_E < sc.par(1)                  # the parameter belongs on the left: src.H_E - par(1) < c
    with pytest.raises(ValueError):
        sc.src.T + sc.src.T + sc.src.T + sc.src.T   # four terms
    bad = sc.Scenario(n_par=1, n_rec=0); p = bad.phase("p"); bad.always(p, [sc.cu("EAS_REF", sc.par(4))])
    with pytest.raises(ValueError):
        bad.pack()


def test_host_interpreter_semantics(fb):
    from flightbatch import scenario as sc
    K = fb.K
    scn = sc.Scenario(n_par=1, n_rec=2)
    A, B, Cc = scn.phase("a"), scn.phase("b"), scn.phase("c")
    scn.always(A, [sc.cu("EAS_REF", 35.0)])
    scn.when(A, sc.src.T >= 0.04, [sc.cu("EAS_REF", sc.cu_("EAS_REF") + 1.0), sc.rec(0, sc.src.T)], then=B)    # sees the `always` action's write
    scn.when(A, sc.ALWAYS, [sc.rec(1, 7.0)], then=Cc)                                                          # never reached behind a rule that fired ...
    scn.always(B, [sc.ui("ENG_START", True)])
    scn.when(B, sc.src.T_IN_PHASE >= 0.04, [sc.ui("ENG_START", False)], then=Cc)
    blob = scn.pack()
    n = 4
    st = _arrays(fb, n); st["par"] = np.zeros((1, n)); st["rec"] = np.full((2, n), np.nan)
    st["active"][3] = False
    dt = 0.02
    for k in range(1, 6):
        st["step"] = k
        sc.evaluate_on_host(blob, st, k * dt, dt)
        if k == 1:   # T = 0.02: the first rule does not hold, the second (always) does: A -> C with rec 1
            assert list(st["phase"][:3]) == [Cc] * 3 and (st["rec"][1, :3] == 7.0).all() and np.isnan(st["rec"][0]).all() and (st["cu"][K["FB_CU_EAS_REF"], :3] == 35.0).all()
            st["phase"][:3] = A   # put them back: now let the first rule fire
        if k == 2:   # T = 0.04: A -> B, ONE transition per evaluation: B's `always` has not run yet
            assert list(st["phase"][:3]) == [B] * 3 and (st["cu"][K["FB_CU_EAS_REF"], :3] == 36.0).all() and (st["rec"][0, :3] == 0.04).all()
            assert (st["ui"][:3] & K["FB_UI_ENG_START"] == 0).all() and (st["since"][:3] == 2).all()
        if k == 3:   # in B: the start bit is set; T_IN_PHASE = 0.02
            assert (st["ui"][:3] & K["FB_UI_ENG_START"] != 0).all() and list(st["phase"][:3]) == [B] * 3
        if k == 4:   # T_IN_PHASE = 0.04: B -> C, bit cleared
            assert list(st["phase"][:3]) == [Cc] * 3 and (st["ui"][:3] & K["FB_UI_ENG_START"] == 0).all()
    assert st["phase"][3] == A and st["cu"][K["FB_CU_EAS_REF"], 3] == 0.0, "an aircraft whose simulation has ended is not evaluated"
