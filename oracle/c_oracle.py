"""ctypes wrapper of oracle/cl_oracle.c (TEST INFRASTRUCTURE, NOT PRODUCT CODE -- see the header of that file).

Builds the parameter / series arrays of the C restatement from a ``DistrictSpec`` (exact float64 device
parameters, not the product's packed float32 tables) and steps ``n_env`` environments.
"""
from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
SRC = HERE / 'cl_oracle.c'
LIB = HERE / 'libcl_oracle.so'

REWARD_KINDS = {'RewardFunction': 0, 'MARL': 1, 'IndependentSACReward': 2, 'SolarPenaltyReward': 3}


def build(force: bool = False) -> Path:
    if force or not LIB.exists() or LIB.stat().st_mtime < SRC.stat().st_mtime:
        subprocess.run(['gcc', '-O2', '-fPIC', '-shared', '-fopenmp', str(SRC), '-o', str(LIB), '-lm'], check=True)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(str(LIB))
        vp = ctypes.c_void_p
        _lib.cl_oracle_step.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp, vp, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_double]
        _lib.cl_oracle_step.restype = None
    return _lib


# slot numbers of cl_oracle.c (kept in sync by test_oracle_golden.py::test_c_oracle_layout)
OP = dict(FLAGS=0, DT=1, R=2, B_CAP=3, B_POW=4, B_LOSS=5, B_CLC=6, B_DOD=7, B_EFF0=8, B_SOC0=9, B_CPC_X=10, B_CPC_Y=13,
          B_PEC_X=16, B_PEC_Y=21, CS=26, HS=32, DS=38, CD_POW=44, CD_EFF=45, CD_TC=46, HD_POW=47, HD_EFF=48, HD_TH=49,
          DD_POW=50, DD_EFF=51, DD_TH=52, DYN_WARMUP=53, ACT_CS=54, ACT_HS=55, ACT_DS=56, ACT_ES=57, ACT_CD=58,
          ACT_HD=59, ACT_COH=60, N=61)
OT = dict(NSL=0, SOLAR_WKW=1, COOL=2, HEAT=3, DHW=4, TOUT=5, PRICE=6, CARBON=7, OUTAGE=8, HVAC=9, PV_POW=10, N=11)
OS = dict(SOC=0, EFF=1, DEGCAP=2, CS=3, HS=4, DS=5, N=6)
OO = dict(NET=0, REWARD=1, EB=2, COOL_DEM=3, C_COOL=4, C_HEAT=5, C_DHW=6, C_NS=7, COST=8, EMISSION=9, N=10)
_ACT = {'cooling_storage': 'ACT_CS', 'heating_storage': 'ACT_HS', 'dhw_storage': 'ACT_DS', 'electrical_storage': 'ACT_ES',
        'cooling_device': 'ACT_CD', 'heating_device': 'ACT_HD', 'cooling_or_heating_device': 'ACT_COH'}


class COracle:
    def __init__(self, spec, tables, n_env: int, reward: str = 'RewardFunction', exponent: float = 1.0,
                 t0_quirk: bool = True):
        L = lib()
        assert (L.cl_oracle_np(), L.cl_oracle_nt(), L.cl_oracle_ns(), L.cl_oracle_no()) == (OP['N'], OT['N'], OS['N'], OO['N'])
        self.L, self.spec, self.n_env = L, spec, n_env
        self.reward, self.exponent, self.t0_quirk = REWARD_KINDS[reward], float(exponent), int(t0_quirk)
        B = len(spec.buildings)
        T = tables.end - tables.start + 1
        self.B, self.T = B, T
        P = np.zeros((B, OP['N']))
        TS = np.zeros((T, B, OT['N']))
        w = slice(tables.start, tables.end + 1)
        col = 0
        for i, b in enumerate(spec.buildings):
            e = b.electrical_storage
            flags = (1 if e.present else 0) | (2 if b.heating_device.is_heat_pump else 0) | (4 if b.dhw_device.is_heat_pump else 0) \
                | (8 if b.outage.simulate else 0) | (16 if b.is_dynamics else 0)
            P[i, OP['FLAGS']] = flags
            P[i, OP['DT']] = b.seconds_per_time_step / 3600
            P[i, OP['R']] = b.time_step_ratio
            P[i, OP['B_CAP']:OP['B_SOC0'] + 1] = [e.capacity, e.nominal_power, e.loss_coefficient, e.capacity_loss_coefficient,
                                                  e.depth_of_discharge, e.efficiency, e.initial_soc]
            P[i, OP['B_CPC_X']:OP['B_CPC_X'] + 3] = e.capacity_power_curve[0]
            P[i, OP['B_CPC_Y']:OP['B_CPC_Y'] + 3] = e.capacity_power_curve[1]
            P[i, OP['B_PEC_X']:OP['B_PEC_X'] + 5] = e.power_efficiency_curve[0]
            P[i, OP['B_PEC_Y']:OP['B_PEC_Y'] + 5] = e.power_efficiency_curve[1]
            for key, tank in (('CS', b.cooling_storage), ('HS', b.heating_storage), ('DS', b.dhw_storage)):
                P[i, OP[key]:OP[key] + 6] = [tank.capacity, tank.loss_coefficient, tank.efficiency, tank.initial_soc,
                                             np.inf if tank.max_input_power is None else tank.max_input_power,
                                             np.inf if tank.max_output_power is None else tank.max_output_power]
            cd, hd, dd = b.cooling_device, b.heating_device, b.dhw_device
            P[i, OP['CD_POW']:OP['CD_TC'] + 1] = [cd.nominal_power, cd.efficiency, cd.target_cooling_temperature]
            P[i, OP['HD_POW']:OP['HD_TH'] + 1] = [hd.nominal_power, hd.efficiency, hd.target_heating_temperature if hd.is_heat_pump else 0]
            P[i, OP['DD_POW']:OP['DD_TH'] + 1] = [dd.nominal_power, dd.efficiency, dd.target_heating_temperature if dd.is_heat_pump else 0]
            P[i, OP['DYN_WARMUP']] = b.dynamics.lookback + 1 if b.dynamics is not None else 0
            for k in _ACT.values():
                P[i, OP[k]] = -1
            for k in b.active_actions:
                P[i, OP[_ACT[k]]] = col
                col += 1
            s = b.series
            TS[:, i, OT['NSL']] = s['non_shiftable_load'][w]
            TS[:, i, OT['SOLAR_WKW']] = s['solar_generation'][w]
            TS[:, i, OT['COOL']] = s['cooling_demand'][w]
            TS[:, i, OT['HEAT']] = s['heating_demand'][w]
            TS[:, i, OT['DHW']] = s['dhw_demand'][w]
            TS[:, i, OT['TOUT']] = s['outdoor_dry_bulb_temperature'][w]
            TS[:, i, OT['PRICE']] = s['electricity_pricing'][w]
            TS[:, i, OT['CARBON']] = s['carbon_intensity'][w]
            TS[:, i, OT['OUTAGE']] = tables.outage[:, i]
            TS[:, i, OT['HVAC']] = s['hvac_mode'][w]
            TS[:, i, OT['PV_POW']] = b.pv_nominal_power
        self.n_act_cols = col
        self.P, self.TS = np.ascontiguousarray(P), np.ascontiguousarray(TS)
        self.state = np.zeros((n_env, B, OS['N']))
        self.out = np.zeros((n_env, B, OO['N']))
        self.out_env = np.zeros((n_env, 4))
        self.reset()

    def reset(self):
        P = self.P
        self.state[:, :, OS['SOC']] = P[:, OP['B_SOC0']].astype(np.float32)
        self.state[:, :, OS['EFF']] = P[:, OP['B_EFF0']]
        self.state[:, :, OS['DEGCAP']] = P[:, OP['B_CAP']]
        for k in ('CS', 'HS', 'DS'):
            self.state[:, :, OS[k]] = P[:, OP[k] + 3].astype(np.float32)
        self.t = 0

    def step(self, actions: np.ndarray, t: int = None):
        """actions float32 [n_act_cols, n_env] (C-contiguous)."""
        t = self.t if t is None else t
        a = np.ascontiguousarray(actions, dtype=np.float32)
        assert a.shape == (self.n_act_cols, self.n_env)
        vp = ctypes.c_void_p
        self.L.cl_oracle_step(self.n_env, self.B, self.P.ctypes.data_as(vp), self.TS.ctypes.data_as(vp),
                              self.state.ctypes.data_as(vp), a.ctypes.data_as(vp), self.out.ctypes.data_as(vp),
                              self.out_env.ctypes.data_as(vp), int(t), self.t0_quirk, self.reward, self.exponent)
        self.t = t + 1
        return self.out, self.out_env
