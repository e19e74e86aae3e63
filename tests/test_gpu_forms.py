"""Every instantiation of collide_kernel that tactics2d_amd/csrc/t2d_collide.hip launches is reached by a recipe of this file --
and each recipe's results are held against the plain form by tests/test_gpu_chain.py / test_gpu_collide.py / test_gpu_envs.py.

The step kernel is ONE body with eight template parameters (DESIGN.md 4.2c lists the forms).  A launch site nobody reaches is code
nobody tests: the debug library notes the template arguments of the last launch (t2d_debug_last_step_kernel, include/t2d_debug.h);
this test drives one pool configuration per site and compares the set it saw with the T2D_LAUNCH_COLLIDE(...) sites in the source.
(Reference: traffic/scenario_manager.py:63-98 is ONE Python loop; the forms are how this build maps it to pool sizes.)"""
import os
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sites():
    """template-argument strings of every launch site of the product build (the -DT2D_EXPERIMENTS block is not compiled)"""
    src = open(os.path.join(ROOT, "tactics2d_amd", "csrc", "t2d_collide.hip")).read()
    src = re.sub(r"#ifdef T2D_EXPERIMENTS.*?#endif", "", src, flags=re.S)
    return sorted(set(re.findall(r"T2D_LAUNCH_COLLIDE\((\([^()]*\)),", src)))


def test_every_launch_site_of_the_step_kernel_is_reached():
    from tactics2d_amd import debug as D, layout as L, scenarios as S
    from tactics2d_amd.controller import IDMController, install
    sites = _sites()
    assert len(sites) == 25, sites
    seen = {}

    def note(what):
        seen.setdefault(D.last_step_kernel(), what)

    def pool_of(sc, variant, idm=False, ego_kernel=True):
        p = D.pool(sc.n_env, sc.A)
        sc.load(p)
        p.set_integrator_variant(variant)
        p.set_auto_reset(True)
        if not ego_kernel:
            p._ck(p._lib.t2d_set_ego_kernel(p._h, 0))
        if idm:
            veh = (sc.rows[sc.type_id, L.P_MODEL] != L.MODEL_POINTMASS).reshape(sc.n_env, sc.A)
            cid = np.full((sc.n_env, sc.A), L.IDM_NONE, np.uint8)
            cid[:, 1:] = np.where(veh[:, 1:], 0, L.IDM_NONE)
            install(p, [IDMController(desired_speed=20.0, horizon=100.0)], cid.reshape(-1))
        a0, a1 = sc.sample_actions(np.random.default_rng(0))
        p.set_actions(a0, a1)
        return p

    mixed_small, highway_small = S.mixed(96, 64, seed=5), S.highway(96, 64, seed=5)
    mixed_big = S.mixed(2200, 64, seed=9)          # 550 workgroups, more than 2 x CUs: the chained form
    parking = S.parking(64, seed0=2)
    for variant in ("exact", "fast"):
        # -- one launch per step
        p = pool_of(highway_small, variant); assert p.step_form(1) == "step"; p.step(100); note("t2d_step, plain pool"); p.close()
        p = pool_of(mixed_small, variant); assert p.step_form(1) == "step_split"; p.step(100); note("t2d_step, one workgroup per env"); p.close()
        p = pool_of(highway_small, variant, idm=True); assert p.step_form(1) == "step"; p.step(100); note("t2d_step, IDM controllers inside"); p.close()
        p = pool_of(parking, variant, ego_kernel=False); p.step(100); note("t2d_step, IoU events on the general kernel"); p.close()
        # -- t2d_step_n
        p = pool_of(mixed_big, variant); assert p.step_form(4) == "chain"; p.step_n(4, 100); note("t2d_step_n, chained"); p.close()
        p = pool_of(mixed_big, variant, idm=True); assert p.step_form(4) == "chain"; p.step_n(4, 100); note("t2d_step_n, chained + IDM"); p.close()
        p = pool_of(mixed_small, variant); p.set_step_chaining(2); assert p.step_form(4) == "chain_split"; p.step_n(4, 100); note("t2d_step_n, chained, one workgroup per env"); p.close()
        p = pool_of(mixed_small, variant); assert p.step_form(4) == "loop_pipe"; p.step_n(4, 100); note("t2d_step_n, PIPE 2 (lane waves)"); p.close()
        p = pool_of(highway_small, variant); assert p.step_form(4) == "loop_pipe"; p.step_n(4, 100); note("t2d_step_n, PIPE 1"); p.close()
        p = pool_of(highway_small, variant, idm=True); assert p.step_form(4) == "loop_pipe"; p.step_n(4, 100); note("t2d_step_n, PIPE 1 + IDM"); p.close()
        p = pool_of(highway_small, variant); p.set_step_chaining(3); assert p.step_form(4) == "loop"; p.step_n(4, 100); note("t2d_step_n, LOOP"); p.close()
    # -- events / status alone (no integrator in the launch: one instantiation per IoU flag)
    p = pool_of(highway_small, "fast"); p.collide(); note("t2d_collide"); p.check_status(100); note("t2d_check_status"); p.close()
    p = pool_of(parking, "fast", ego_kernel=False); p.check_status(100); note("t2d_check_status with IoU events"); p.close()
    missing = [s for s in sites if s not in seen]
    extra = [s for s in seen if s not in sites]
    assert not missing and not extra, dict(missing=missing, extra=extra, seen=seen)
