/*
 * citylearn_amd.h -- C-ABI of the MI355X-native CityLearn step engine.
 *
 * The reference (intelligent-environments-lab/CityLearn v2.4.2) is pure Python and has no FFI; its
 * boundary for this path is `CityLearnEnv.step` (citylearn/citylearn.py:978-1056) which loops
 * `Building.apply_actions` (citylearn/building.py:1500-1634), `Building.update_variables`
 * (building.py:2615-2703), `RewardFunction.calculate` (citylearn/reward_function.py:65-88 and subclasses)
 * and `Building.next_time_step` (building.py:2502-2524) over the buildings of ONE district.  The entry
 * points below are what a ctypes binding inside the reference would call instead of those loops, for a
 * whole batch of independent districts ("envs") at once.  See INTEGRATION.md for the reference-side stub.
 *
 * Conventions
 *   - every pointer is a caller-owned DEVICE buffer (HBM); the library never allocates, frees or retains it;
 *   - all entry points are asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - return value 0 = success, negative CL_E* otherwise; `cl_last_error()` gives a thread-local message;
 *   - no C++ types, no torch types, no exceptions cross the boundary;
 *   - base pointers must be 16-byte aligned; `n_env` must be a multiple of 4 (pad the batch).
 *
 * Memory layout (structure of arrays, env index fastest => coalesced 64-lane wavefront access)
 *   params  [n_bldg][CL_NP]            32-bit words, per-building static device parameters (cl_param slots)
 *   ts      [n_steps][n_bldg][CL_NF]   f32, per-(time step, building) time-series row (cl_feat slots);
 *                                      identical for every env (all envs replay the same episode window)
 *   state   [CL_NS][n_bldg][n_env]     f32, carried device state (cl_state planes)
 *   actions element (col, env) at actions[col*act_stride_col + env*act_stride_env]; one column per active
 *                                      (building, action) pair in the reference's central-agent order
 *                                      (citylearn.py:1069-1079); [n_act_cols][n_env] is the coalesced layout
 *   out_bldg [CL_NO][n_bldg][n_env]    f32, per-building outputs of the step (cl_out planes)
 *   out_env  [CL_NQ][n_env]            f32, district sums over buildings (cl_envout planes)
 *   kpi_bldg [CL_NKB][n_bldg][n_env], kpi_env [CL_NKE][n_env]   optional streaming KPI accumulators
 * Adjacent stages keep their own tables and planes, described at their entry points below: the LSTM indoor-temperature
 * stage (cl_lstm_*), the observation epilogue (cl_observe_f32), flexible loads = EV chargers / EVs / washing machines (cl_flex).
 */
#ifndef CITYLEARN_AMD_H
#define CITYLEARN_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CL_ABI_VERSION 8   /* 8: CLD_CHECK (the reference's runtime assertions as a violation word per unit), tickets of the in-launch fold at the reserved plane's tail; 7: CLD_F64_CHAIN (params rows of 320 words: CLP_C_* block); 6: cl_finish_f32, cl_tuning.finish = 3 (deferred finish); 5: cl_tuning.kernel_name, CLD_F64_MAPS; 4: LSTM tables with pre-scaled gate rows, CLD_LSTM_F16 (two-term f16 `lstm_wb`) */

/* ---- error codes ---- */
#define CL_OK            0
#define CL_EINVAL       -1   /* bad dims / flag combination */
#define CL_ENULL        -2   /* required pointer is NULL */
#define CL_EALIGN       -3   /* pointer or n_env alignment */
#define CL_EHIP         -4   /* HIP runtime error (message in cl_last_error) */
#define CL_ERANGE       -5   /* t / k_steps outside [0, n_steps) */

/* ---- table widths ---- */
#define CL_NP  320   /* words per building in `params` (1.25 KiB rows) */
#define CL_NF   16   /* floats per (t, building) row in `ts` */
#define CL_NS    8   /* state planes (the last two are only touched under CLD_F64_MAPS) */
#define CL_NO   18   /* per-building output planes */
#define CL_NQ    4   /* per-env (district) output planes */
#define CL_NKB  12   /* per-building KPI accumulator planes */
#define CL_NKE  24   /* per-env KPI accumulator planes */

/* ---- per-building parameter slots (`params[b][slot]`, f32 unless noted) ---- */
enum cl_param {
    CLP_FLAGS = 0,        /* u32 bit pattern, CLF_* below */
    CLP_DT_HOURS,         /* seconds_per_time_step/3600 (building.py:113) */
    CLP_TSR,              /* time_step_ratio r (data.py:427-455); 1.0 for hourly data + hourly control */
    /* electrical storage (energy_model.py:872-1141) */
    CLP_B_CAP, CLP_B_POW, CLP_B_LOSS /* loss_coefficient*r */, CLP_B_CLC, CLP_B_DOD, CLP_B_EFF0, CLP_B_SOC0,
    CLP_B_CPC_X0, CLP_B_CPC_X1, CLP_B_CPC_X2,          /* capacity_power_curve soc breakpoints */
    CLP_B_CPC_Y0, CLP_B_CPC_Y1, CLP_B_CPC_Y2,          /* ... power fractions */
    CLP_B_PEC_X0, CLP_B_PEC_X1, CLP_B_PEC_X2, CLP_B_PEC_X3, CLP_B_PEC_X4,   /* power_efficiency_curve */
    CLP_B_PEC_Y0, CLP_B_PEC_Y1, CLP_B_PEC_Y2, CLP_B_PEC_Y3, CLP_B_PEC_Y4,
    /* thermal storages (energy_model.py:603-870): capacity, loss*r, sqrt(efficiency), initial soc, max in / out */
    CLP_CS_CAP, CLP_CS_LOSS, CLP_CS_RTE, CLP_CS_SOC0, CLP_CS_MAXIN, CLP_CS_MAXOUT,
    CLP_HS_CAP, CLP_HS_LOSS, CLP_HS_RTE, CLP_HS_SOC0, CLP_HS_MAXIN, CLP_HS_MAXOUT,
    CLP_DS_CAP, CLP_DS_LOSS, CLP_DS_RTE, CLP_DS_SOC0, CLP_DS_MAXIN, CLP_DS_MAXOUT,
    /* electric devices: nominal power [kW] (energy_model.py:85-155) */
    CLP_CD_POW, CLP_HD_POW, CLP_DD_POW,
    /* divisor used by the reference for the t=0 heating re-add when heating_device is an ElectricHeater
       (building.py:2632 uses dhw_device.get_input_power, SURVEY App.B6) */
    CLP_T0_HEAT_DIV,
    /* heat-pump / heater nameplate (only read by the oracle; the kernel reads precomputed COP columns in `ts`) */
    CLP_CD_EFF, CLP_CD_TC, CLP_HD_EFF, CLP_HD_TH, CLP_DD_EFF, CLP_DD_TH,
    /* LSTMDynamicsBuilding: first step index at which the device action overrides the ideal load
       (lookback + 1; building.py:2998-2999, 3108) */
    CLP_DYN_WARMUP,
    /* action columns (i32 bit pattern; -1 = action inactive for this building; building.py:1557-1564) */
    CLP_ACT_COOL_STO, CLP_ACT_HEAT_STO, CLP_ACT_DHW_STO, CLP_ACT_ELEC_STO,
    CLP_ACT_COOL_DEV, CLP_ACT_HEAT_DEV, CLP_ACT_COH_DEV,
    /* reward parameters (reward_function.py) */
    CLP_RW_EXPONENT,      /* RewardFunction.exponent */
    /* ---- derived values, computed by the host packer in float64 and rounded once (the kernels read these
     *      instead of dividing by wave-uniform quantities).  The "lean" battery+PV kernel reads only the
     *      contiguous block CLP_L_FIRST .. CLP_L_LAST (two 16-dword scalar loads). ---- */
    CLP_L_FIRST = 64,
    CLP_L_FLAGS = CLP_L_FIRST, /* copy of CLP_FLAGS */
    CLP_L_ACT_ES,         /* copy of CLP_ACT_ELEC_STO */
    CLP_L_TSR,            /* r */
    CLP_L_PDT,            /* nominal_power * dt  [kWh per unit action] */
    CLP_L_POW,            /* nominal_power */
    CLP_L_CAP,            /* capacity */
    CLP_L_CAPL,           /* capacity * (1 - loss*r) */
    CLP_L_INV_CAP,        /* 1 / max(capacity, ZDP) */
    CLP_L_INV_POW,        /* 1 / max(nominal_power, ZDP) */
    CLP_L_OMD,            /* 1 - depth_of_discharge */
    CLP_L_DEGK,           /* capacity_loss_coefficient * capacity * r / 2 */
    CLP_L_CPC_X1,         /* capacity_power_curve breakpoint; pmax = A + B*soc on either side (already times P) */
    CLP_L_CPC_A0, CLP_L_CPC_B0, CLP_L_CPC_A1, CLP_L_CPC_B1,
    CLP_L_PEC_X1, CLP_L_PEC_X2, CLP_L_PEC_X3,          /* power_efficiency_curve: eff = A + B*x per segment */
    CLP_L_PEC_A0, CLP_L_PEC_B0, CLP_L_PEC_A1, CLP_L_PEC_B1, CLP_L_PEC_A2, CLP_L_PEC_B2, CLP_L_PEC_A3, CLP_L_PEC_B3,
    CLP_L_RW_EXPONENT,    /* copy of CLP_RW_EXPONENT */
    CLP_L_SOC0, CLP_L_EFF0,   /* copies of CLP_B_SOC0 / CLP_B_EFF0 */
    CLP_L_LAST = CLP_L_EFF0,
    /* tanks: 1/sqrt(eff), 1/max(cap, ZDP), cap*(1-loss*r) */
    CLP_CS_IRTE, CLP_CS_ICAP, CLP_CS_CAPL,
    CLP_HS_IRTE, CLP_HS_ICAP, CLP_HS_CAPL,
    CLP_DS_IRTE, CLP_DS_ICAP, CLP_DS_CAPL,
    CLP_T0_IHEAT_DIV,     /* 1 / CLP_T0_HEAT_DIV */
    CLP_FLEX_INDEX,       /* i32: row of this building in the flexible-load planes (cl_flex.flex_out), -1 = no charger / washing machine */
    CLP_USED,
    /* ---- compact copy for the thermal / outage step kernel (cl_full.h): 64 words = four 64-byte lines, in the order the kernel
     *      consumes them, so that a wave fetches a building's parameters in three batches of independent scalar loads instead of a
     *      dozen dependent round trips scattered over the slots above.  Same bit patterns as the slots they copy. ---- */
    CLP_F_FIRST = 128,
    CLP_F_FLAGS = CLP_F_FIRST,                       /* CLP_FLAGS, then the seven action columns CLP_ACT_COOL_STO .. CLP_ACT_COH_DEV */
    CLP_F_ACT = CLP_F_FIRST + 1,
    CLP_F_HEAD = CLP_F_FIRST + 8,                    /* dt, r, cooling / heating / dhw device power, CLP_T0_IHEAT_DIV, CLP_DYN_WARMUP, reward exponent */
    CLP_F_TANK = CLP_F_FIRST + 16,                   /* 3 x 8 (cooling, heating, dhw): capacity, capacity*(1-loss r), sqrt(eff), 1/sqrt(eff), 1/capacity,
                                                        max input, max output, action scale [kWh per unit action] (sic: building.py:1676, 1720, 1765) */
    CLP_F_BATT = CLP_F_FIRST + 40,                   /* the 24 words CLP_L_PDT .. CLP_L_PEC_B3 */
    CLP_F_LAST = CLP_F_FIRST + 63,
    /* ---- CLD_F64_MAPS: the battery's parameters as float64 (two words each, little endian; cl_param_f64 indexes them), unrounded:
     *      the reference computes Battery.charge mostly in float64 (energy_model.py:1027-1141; csrc/cl_unit.h battery_charge_ref). ---- */
    CLP_D_FIRST = 192,
    CLP_D_LAST = CLP_D_FIRST + 63,
    /* ---- CLD_F64_CHAIN: the battery map's constants as float64 in the form the fast float64 chain consumes them (cl_param_chain indexes
     *      them; csrc/cl_unit.h battery_charge_chain): both curves as a first segment plus one ramp per further breakpoint. ---- */
    CLP_C_FIRST = 256,
    CLP_C_LAST = CLP_C_FIRST + 63
};
enum cl_param_chain {     /* k-th double of the CLP_C_* block */
    CLPC_CAP = 0,         /* capacity */
    CLPC_OML,             /* 1 - loss_coefficient * r */
    CLPC_RCAP,            /* 1 / max(capacity, ZERO_DIVISION_PLACEHOLDER) */
    CLPC_PDT,             /* nominal_power * seconds_per_time_step / 3600  [kWh per unit action] */
    CLPC_POW, CLPC_RPOW,  /* nominal_power, 1 / max(nominal_power, ZERO_DIVISION_PLACEHOLDER) */
    CLPC_TSR,             /* time_step_ratio r */
    /* capacity_power_curve (energy_model.py:1070-1090), times nominal_power: pmax = A0 + B0 soc + DB1 max(soc - X1, 0) */
    CLPC_CPC_A0, CLPC_CPC_B0, CLPC_CPC_X1, CLPC_CPC_DB1,
    /* power_efficiency_curve (energy_model.py:1092-1109): eff = A0 + B0 x + sum_k DBk max(x - Xk, 0) */
    CLPC_PEC_A0, CLPC_PEC_B0, CLPC_PEC_X1, CLPC_PEC_DB1, CLPC_PEC_X2, CLPC_PEC_DB2, CLPC_PEC_X3, CLPC_PEC_DB3,
    CLPC_VALID,           /* 1.0 when the curves have the shape the ramp form assumes (breakpoints ascending, the last one >= 1, power fractions
                             <= 1: the reference's out-of-range rule -- segment 0 again beyond the last breakpoint -- is then unreachable), else 0.0 */
    CLPC_USED             /* <= 32 */
};
enum cl_param_f64 {       /* k-th double of the CLP_D_* block */
    CLPD_TSR = 0,         /* time_step_ratio r */
    CLPD_DT,              /* seconds_per_time_step / 3600 */
    CLPD_POW, CLPD_CAP,   /* nominal_power, capacity */
    CLPD_OML,             /* 1 - loss_coefficient * r */
    CLPD_SOC_LIMIT,       /* 1 - depth_of_discharge */
    CLPD_CLCCAP,          /* capacity_loss_coefficient * capacity */
    CLPD_EFF0,            /* Battery.efficiency at reset */
    CLPD_CPC_X0, CLPD_CPC_X1, CLPD_CPC_X2, CLPD_CPC_Y0, CLPD_CPC_Y1, CLPD_CPC_Y2,          /* capacity_power_curve (x: soc, y: power fraction) */
    CLPD_PEC_X0, CLPD_PEC_X1, CLPD_PEC_X2, CLPD_PEC_X3, CLPD_PEC_X4,                       /* power_efficiency_curve */
    CLPD_PEC_Y0, CLPD_PEC_Y1, CLPD_PEC_Y2, CLPD_PEC_Y3, CLPD_PEC_Y4,
    /* correctly rounded float64 reciprocals of the map's CONSTANT divisors (round 4): battery_charge_ref divides by them through two
       fused-multiply-add correction steps (Markstein), which returns the correctly rounded quotient -- the same bits as the division the
       reference executes -- in 5 FMAs instead of the ~11-instruction v_div_scale / v_rcp_f64 / v_div_fmas sequence */
    CLPD_RCAP,            /* 1 / max(capacity, ZERO_DIVISION_PLACEHOLDER) */
    CLPD_RPOW,            /* 1 / max(nominal_power, ZERO_DIVISION_PLACEHOLDER) */
    CLPD_RCPC_01, CLPD_RCPC_12,                             /* 1 / (cpc_x[k + 1] - cpc_x[k]) */
    CLPD_RPEC_01, CLPD_RPEC_12, CLPD_RPEC_23, CLPD_RPEC_34, /* 1 / (pec_x[k + 1] - pec_x[k]) */
    CLPD_USED             /* <= 32 */
};

/* ---- building flag bits (CLP_FLAGS) ---- */
#define CLF_BATTERY      (1u << 0)   /* electrical_storage present */
#define CLF_COOL_DEV     (1u << 1)
#define CLF_HEAT_DEV     (1u << 2)
#define CLF_DHW_DEV      (1u << 3)
#define CLF_COOL_STO     (1u << 4)
#define CLF_HEAT_STO     (1u << 5)
#define CLF_DHW_STO      (1u << 6)
#define CLF_HEAT_IS_HP   (1u << 7)   /* heating_device is a HeatPump (else ElectricHeater) */
#define CLF_DHW_IS_HP    (1u << 8)
#define CLF_OUTAGE       (1u << 9)   /* simulate_power_outage (building.py:671-674) */
#define CLF_DYNAMICS     (1u << 10)  /* LSTMDynamicsBuilding: partial-load cooling/heating demand (building.py:3080-3158) */
#define CLF_FLEX         (1u << 11)  /* building has EV chargers and / or washing machines (cl_flex; CLP_FLEX_INDEX >= 0) */
#define CLF_THERMAL      (CLF_COOL_DEV | CLF_HEAT_DEV | CLF_DHW_DEV | CLF_COOL_STO | CLF_HEAT_STO | CLF_DHW_STO)

/* ---- time-series row (`ts[t][b][feat]`) ---- */
enum cl_feat {
    CLT_NSL = 0,      /* non_shiftable_load [kWh] */
    CLT_SOLAR,        /* -pv.nominal_power * W_per_kW / 1000  (<= 0; building.py:2554) */
    CLT_COOL_DEM, CLT_HEAT_DEM, CLT_DHW_DEM,           /* ideal demands [kWh] */
    CLT_COP_COOL, CLT_COP_HEAT, CLT_COP_DHW,           /* HeatPump.get_cop (energy_model.py:216-250) or heater efficiency */
    CLT_PRICE, CLT_CARBON,                             /* electricity_pricing, carbon_intensity */
    CLT_OUTAGE,       /* power-outage signal 0/1 (power_outage.py:131-169), already AND-ed with simulate_power_outage */
    CLT_HVAC_MODE,    /* 0 off, 1 cooling, 2 heating, 3 auto (data.py:341) */
    CLT_T_OUT,        /* outdoor_dry_bulb_temperature [C] (oracle recomputes COP from it) */
    CLT_ICOP_COOL, CLT_ICOP_HEAT, CLT_ICOP_DHW          /* reciprocals of the three COP columns */
};

/* ---- state planes (`state[plane][b][env]`) ---- */
enum cl_state {
    CLS_B_SOC = 0,    /* electrical_storage.soc[t] */
    CLS_B_EFF,        /* Battery.efficiency left by the previous charge() call (energy_model.py:1039-1052) */
    CLS_B_DEGCAP,     /* Battery.degraded_capacity [kWh] (under CLD_F64_CHAIN: capacity - degraded_capacity, the accumulated loss) */
    CLS_CS_SOC, CLS_HS_SOC, CLS_DS_SOC,                /* cooling / heating / dhw tank soc[t] */
    CLS_B_EFF_LO, CLS_B_DEGCAP_LO                      /* CLD_F64_MAPS: low words of the two float64 values the reference carries between steps --
                                                          Battery.efficiency = (double)CLS_B_EFF + (double)CLS_B_EFF_LO, likewise the degraded
                                                          capacity; written by cl_reset_f32, left alone by the fp32 kernels */
};

/* ---- per-building outputs (`out_bldg[plane][b][env]`) ---- */
enum cl_out {
    CLO_NET = 0,      /* net_electricity_consumption[t] (building.py:2681-2694) */
    CLO_REWARD,       /* per-building reward (reward_function.py) */
    CLO_B_EB,         /* electrical_storage.energy_balance[t] */
    CLO_COOL_DEM,     /* delivered cooling: energy_from_cooling_device + |min(eb_cs,0)| (building.py:1435-1437) */
    CLO_C_COOL, CLO_C_HEAT, CLO_C_DHW, CLO_C_NSL,      /* device electricity_consumption[t] */
    CLO_BASE_NET,     /* baseline net of evaluate(): net_electricity_consumption_without_storage (building.py:345-366)
                         or ..._and_partial_load for dynamics buildings (building.py:2877-2905) */
    CLO_EXPECTED,     /* cooling + heating + dhw demand + non_shiftable_load (citylearn.py:1216) */
    CLO_SERVED,       /* energy from devices + storages + energy_to_non_shiftable_load (citylearn.py:1217-1220) */
    CLO_HEAT_DEM,     /* delivered heating: energy_from_heating_device + |min(eb_hs,0)| (building.py:1436) */
    CLO_DHW_DEM,      /* delivered dhw:     energy_from_dhw_device + |min(eb_ds,0)|     (building.py:1437) */
    CLO_NET_WS,       /* net_electricity_consumption_without_storage (building.py:345-366): equals CLO_BASE_NET except for dynamics
                         buildings, whose default baseline also removes the partial-load difference (evaluate()'s
                         EvaluationCondition variants, citylearn.py:29-50) */
    CLO_SE_COOL, CLO_SE_HEAT, CLO_SE_DHW,   /* cooling / heating / dhw _storage_electricity_consumption[t]: the tank's energy balance through the
                         device's COP or efficiency (building.py:413-457); detail planes */
    CLO_RESERVED      /* scratch of building-chunked launches (per-chunk district partial sums, then one arrival counter per env tile; with
                         cl_tuning.finish = 3 two such row sets, double-buffered by step parity, and three marker words in the plane's last
                         16 bytes); keep last.  The caller zero-fills `out_bldg` once before the first step: the counters return to zero by
                         themselves */
};

/* ---- district outputs (`out_env[plane][env]`) ---- */
enum cl_envout {
    CLQ_NET = 0, CLQ_COST, CLQ_EMISSION,               /* citylearn.py:1909-1918 */
    CLQ_REWARD                                         /* sum over buildings (central_agent reward) */
};

/* ---- streaming KPI accumulators (CLD_KPI; what CityLearnEnv.evaluate needs, citylearn.py:1136-1323) ---- */
enum cl_kpi_bldg {            /* kpi_bldg[plane][b][env]: running sums over the episode */
    CLK_C_POS = 0,            /* sum max(net, 0)            -> electricity_consumption_total (cost_function.py:114) */
    CLK_C_NET,                /* sum net                    -> zero_net_energy               (cost_function.py:136) */
    CLK_C_EMISSION,           /* sum max(emission, 0)       -> carbon_emissions_total        (cost_function.py:159) */
    CLK_C_COST,               /* sum max(cost, 0)           -> cost_total                    (cost_function.py:179) */
    CLK_B_POS, CLK_B_NET, CLK_B_EMISSION, CLK_B_COST,   /* the same on the baseline (no-storage) series */
    CLK_UNSERVED_OUTAGE, CLK_EXPECTED_OUTAGE,            /* normalized_unserved_energy, outage steps (cost_function.py:356) */
    CLK_UNSERVED_ALL, CLK_EXPECTED_ALL                   /* ... all steps */
};
enum cl_kpi_env {             /* kpi_env[cond*12 + k][env], cond 0 = control district net, 1 = baseline district net */
    CLKE_PREV = 0,            /* previous value (for ramping, cost_function.py:10) */
    CLKE_RAMP,                /* sum of positive first differences */
    CLKE_DAY_SUM, CLKE_DAY_MAX,      /* open 24-step group */
    CLKE_DAY_LF_SUM,          /* sum over closed days of 1 - mean/max (cost_function.py:62) */
    CLKE_DAY_PEAK_SUM,        /* sum over closed days of the daily max (cost_function.py:89) */
    CLKE_DAY_N,               /* closed days */
    CLKE_MON_SUM, CLKE_MON_MAX, CLKE_MON_LF_SUM, CLKE_MON_N,   /* the same for 730-step groups */
    CLKE_ALL_MAX,             /* all-time peak */
    CLKE_PER_COND
};

/* ---- step flags (`cl_dims.flags`) ---- */
#define CL_ROW0_BLOCK 256   /* envs per episode-offset block (cl_dims.env_row0); every kernel's env tile divides it */

#define CLD_REF_T0_QUIRK   (1u << 0)  /* replicate the reference's repeated t=0 update_variables (SURVEY App.B1) */
#define CLD_WRITE_DETAIL   (1u << 1)  /* also write CLO_B_EB .. CLO_C_NSL planes (parity / KPI baselines) */
#define CLD_KPI            (1u << 2)  /* update the streaming KPI accumulators.  The step launch updates them itself -- no detail planes needed -- for
                                         thermal / outage districts of up to 32 buildings without flexible loads and without CLD_F64_MAPS
                                         (cl_step_full_kpi_kernel) and for CLD_LEAN districts of up to 32 buildings stepped without flexible loads;
                                         every other district needs CLD_WRITE_DETAIL (a launch after the step reads the planes).  In such a CLD_LEAN
                                         district the baseline (net without the battery = load + solar), the expected energy and the baseline
                                         district series do not depend on the env, so without CLD_WRITE_DETAIL the planes CLK_B_POS .. CLK_B_COST,
                                         CLK_EXPECTED_ALL and the condition-1 rows of `kpi_env` are maintained ONCE per block of CL_ROW0_BLOCK
                                         envs, at the block's first env -- the other entries keep their reset values -- and per (env, building)
                                         only the four control sums move: 32 B next to the step's 37) */
#define CLD_ES_COL_IS_BLDG   (1u << 4)  /* hint: the electrical_storage action column of building b is column b (one action per
                                          building, building order) -- lets the step issue its action loads before the parameters */
#define CLD_CENTRAL_AGENT  (1u << 5)  /* CLR_EV only: central_agent districts scale every building's charger terms by the DISTRICT
                                         MARL reward (reward_function.py:423-425 takes current_reward[0], the sum) */
#define CLD_LEAN           (1u << 3)  /* caller asserts: no building has a thermal device / tank, outage or dynamics
                                         flag (battery + PV + non-shiftable load only, e.g. the 2022 schemas) ->
                                         the specialised lean kernel may be used */
#define CLD_LSTM_F16       (1u << 6)  /* cl_lstm_step_f32 only: `lstm_wb` holds two f16 terms per weight (dynamics.pack_lstm_split(.., 'f16'))
                                         instead of three bf16 terms */
#define CLD_F64_MAPS       (1u << 7)  /* cl_step_f32 / cl_step_flex_f32: evaluate the battery map in float64 on the CLP_D_* parameters, round soc[t] and
                                         energy_balance[t] to float32 where the reference's float32 series do, carry efficiency / degraded capacity
                                         as hi + lo planes -- the reference's own precision model, for free-running parity at 1e-4 (the default
                                         fp32 map is locally expansive on the steep part of the capacity-power curve: DESIGN.md section 3).
                                         Slower launches (general / lean step kernels only; not the fused rollout, the env-major or the
                                         thermal-specialised kernels). */
#define CLD_F64_CHAIN      (1u << 14) /* cl_step_f32 / cl_rollout_f32: the battery's soc chain -- energy_init, capacity-power limit, efficiency, final energy, soc[t],
                                         energy_balance[t] (energy_model.py:1027-1109, 719-768) -- in float64 on the CLP_C_* constants, and the degraded
                                         capacity carried as the LOSS `capacity - degraded_capacity` in the CLS_B_DEGCAP plane (float32 holds it to
                                         ~2^-40 of the capacity; cl_reset_f32 writes 0 there under this flag).  Not bit-identical to the reference
                                         like CLD_F64_MAPS, but free-running inside 1e-4 on every fixture at a fraction of its cost and with the
                                         default three state planes: what seeds the fp32 map's drift on the steep part of the capacity-power
                                         curve is the float32 rounding of the degraded capacity, not the arithmetic.
                                         Available in every step kernel and in the fused rollout; mutually exclusive with CLD_F64_MAPS;
                                         districts with flexible loads keep the fp32 map (their EV batteries are fp32).  With CLD_KPI:
                                         battery + PV districts of up to 32 buildings update the accumulators inside the step launch as in
                                         fp32 (cl_step_lean_kpi_chain_kernel); thermal / chunked districts need CLD_WRITE_DETAIL (the KPI
                                         launch reads the detail subset).  cl_step_observe_f32 fills the observation tile from the same
                                         launch as in fp32 (cl_step_lean_obs_chain_kernel). */
#define CLD_DETAIL_MIN     (1u << 12) /* with CLD_WRITE_DETAIL: write only the detail planes another kernel of the path reads -- CLO_BASE_NET,
                                         CLO_EXPECTED, CLO_SERVED (the streaming KPI pass) and CLO_COOL_DEM, CLO_HEAT_DEM (the LSTM stage) -- and
                                         leave the other ten alone (5 instead of 15 extra planes per step) */
#define CLD_LSTM_TWO_DEMANDS (1u << 13) /* cl_lstm_step_f32 only: caller asserts that a building of the district has a temperature model taking BOTH
                                          demands (lstm_w[CLW_DEM2] != 0; needs `heat_dem`): selects the instantiation that reads the third input
                                          ring (rows 24-35 of `hist`).  Without it such a building's indoor_temp is NaN */
#define CLD_CHECK          (1u << 15) /* debug mode of cl_step_f32 / cl_step_flex_f32 (ABI 8): evaluate the reference's own runtime assertions inside the
                                         step and leave one word of CLV_* bits per (building, env) in plane CLO_RESERVED of `out_bldg` (read it as
                                         uint32; 0 = the reference would not have raised).  Needs CLD_WRITE_DETAIL and a district the launch does not
                                         cut into building chunks (n_bldg <= 32), whose scratch the plane otherwise is; selects the general step kernel
                                         (cl_step_kernel<1, true, true, .., CHECK = true>), one env per lane.  The demand-limit assertion
                                         (building.py:1825-1829) only involves env-independent operands and stays a host table (CityLearnEnv). */
enum cl_violation {
    CLV_FLEXIBILITY = 1,   /* downward_electrical_flexibility < 0 beyond TOLERANCE during a power outage (building.py:665) */
    CLV_COOLING     = 2,   /* ___electricity_consumption_polarity_check('cooling', ..): negative device consumption (building.py:1660, 1831-1835) */
    CLV_HEATING     = 4,   /* ... 'heating' (building.py:1708) */
    CLV_DHW         = 8,   /* ... 'dhw' (building.py:1753) */
    CLV_NSL         = 16   /* ElectricDevice.update_electricity_consumption(enforce_polarity): a negative non-shiftable load (energy_model.py:146-148) */
};
#define CLD_REWARD_SHIFT   8          /* reward kind in bits 8..11 */
#define CLD_REWARD_MASK    (0xFu << CLD_REWARD_SHIFT)
enum cl_reward_kind {
    CLR_DEFAULT = 0,          /* RewardFunction: -max(net,0)**exponent          (reward_function.py:65-88)  */
    CLR_MARL = 1,             /* MARL                                            (reward_function.py:132-143) */
    CLR_INDEPENDENT_SAC = 2,  /* IndependentSACReward: min(-net, 0)             (reward_function.py:159-168) */
    CLR_SOLAR_PENALTY = 3,    /* SolarPenaltyReward                              (reward_function.py:189-214) */
    CLR_EV = 4                /* Electric_Vehicles_Reward_Function (needs cl_flex)  (reward_function.py:389-531) */
};

/* Launch-geometry overrides for tests and tuning scripts (cl_dims.tuning).  Every field 0 = the library's own choice.
 * They travel with the call: the library keeps no mutable state of its own (re-entrant, thread-safe for disjoint buffers).
 * Results never depend on them beyond the documented last-bit summation-order effects of the env-major kernels. */
typedef struct cl_tuning {
    int32_t vec;            /* envs per lane of the step / rollout kernels: 1, 2 or 4 (the env-major kernel: 2 = two envs per lane, measured slower) */
    int32_t nw;             /* waves per workgroup (= building lanes) */
    int32_t no_chunks;      /* 1: never cut the building axis into gridDim.y chunks */
    int32_t lean_variant;   /* lean districts, bit mask: 1 = general kernel, 2 = latency-ordered lean kernel at any grid size, 4 = through the thermal
                               kernel (experiments), 8 = the env-major kernel's general 20-building instantiation where the 17-building one would run (A/B),
                               16 = building-chunked launches through cl_step_kernel instead of cl_step_lean_chunk_kernel (tests, A/B) */
    int32_t envmajor;       /* env-major kernels (one lane = one env x all buildings): 0 = by batch size, 1 = always, 2 = never */
    int32_t flex_vec;       /* envs per lane of the flexible-load kernel: 1, 2 or 4 */
    int32_t obs_variant;    /* observation epilogue: 1 row-wise, 2 LDS-tile, 3 wave-independent, 4 plane-transpose kernel (all columns env-dependent),
                               5 row-wise with the dependent-column list in the kernel arguments (one round trip; default for wide vectors) */
    int32_t obs_rows;       /* LDS-tile observation kernel: envs per block */
    int32_t lstm_variant;   /* LSTM stage timing experiments (csrc/cl_lstm.h) */
    int32_t full_variant;   /* thermal / outage districts (tests, A/B): 1 = the round-1 general kernel; 2 = parameter blocks staged in LDS even when
                               the launch is not building-chunked; 3 = one env tile per workgroup, parameters in SGPRs (neither LDS staging nor
                               the multi-tile kernel); 5 = the multi-tile kernel (vec = envs per lane, nw = waves, b_chunk = tiles per workgroup) */
    int32_t b_chunk;        /* building-chunked launches: buildings per workgroup row (with `nw` waves per workgroup) */
    int32_t nt_stores;      /* non-temporal hint on the step kernels' plane stores: 0 = by launch footprint, 1 = always, 2 = never */
    char* kernel_name;      /* nullable HOST buffer of CL_KERNEL_NAME_LEN bytes: cl_step_f32 / cl_step_flex_f32 / cl_step_observe_f32 /
                               cl_rollout_f32 / cl_lstm_step_f32 write the instantiation(s) they launched into it, '+'-separated, in the
                               spelling rocprofv3 prints (bench.py's `roofline.kernel`, scripts/profile_round.sh's name check).  Output only:
                               nothing the library computes depends on it. */
    int32_t finish;         /* building-chunked launches (districts of more than 32 buildings): 0 / 1 = a second launch folds the chunk partial
                               sums (cl_finish_kernel), 2 = the last chunk of an env tile to arrive folds them inside the step launch
                               (measured slower: csrc/cl_kernels.hip district_reduce; tests, A/B); 3 = DEFERRED: the step launch folds the
                               PREVIOUS step's chunk sums and leaves its own in the scratch rows -- `out_env` then trails the step by one
                               launch until cl_finish_f32 (below) is called.  Only for rewards that do not couple the buildings (not
                               MARL / EV), without CLD_KPI / CLD_F64_MAPS / CLD_WRITE_DETAIL / flexible loads (CLD_F64_CHAIN: battery + PV districts only); every other call keeps
                               the second launch, and cl_finish_f32 is then a no-op (a launch that does not defer clears its step parity's marker, so the mode
                               may change between steps on live buffers).  cl_rollout_seq_f32 finishes its last step itself. */
    int32_t kpi_passes;     /* streaming KPIs of thermal / outage districts and of districts stepped with the detail planes: 0 = inside the step
                               launch where the launch is the one-env-per-lane thermal kernel (cl_step_full_kpi_kernel), else one launch after
                               the step; 1 = always the launch after the step (cl_kpi_kernel; needs CLD_WRITE_DETAIL), 2 = the two passes of
                               rounds 1 - 2 (cl_kpi_bldg_kernel + cl_kpi_env_kernel; tests, A/B) */
} cl_tuning;
#define CL_KERNEL_NAME_LEN 256

typedef struct cl_dims {
    int32_t n_env;        /* envs in this shard (multiple of 4) */
    int32_t n_bldg;       /* buildings per district */
    int32_t n_steps;      /* episode_time_steps: valid t are 0 .. n_steps-1 */
    int32_t n_act_cols;   /* action columns */
    uint32_t flags;       /* CLD_* */
    int32_t n_ts_rows;    /* rows in `ts` / `dyn_pre` / `obs_table`; 0 means n_steps */
    const int32_t* env_row0;  /* nullable DEVICE pointer [ceil(n_env / CL_ROW0_BLOCK)]: per-env-block episode offset.
                                 Env block g (CL_ROW0_BLOCK consecutive envs) reads table row env_row0[g] + t at step t, so
                                 different blocks replay different windows of the simulation period at once (the batched
                                 analogue of EpisodeTracker's rolling / random episode splits, base.py:100-129).  The caller
                                 guarantees 0 <= env_row0[g] and env_row0[g] + n_steps <= n_ts_rows. */
    const cl_tuning* tuning;  /* nullable HOST pointer, read during the call only */
    int64_t env_offset;       /* index of this shard's first env in the whole (multi-GPU) batch: added to the env index wherever it keys
                                 a random stream (rollout policy, unconnected-EV drift), so that ranks given the same seed draw disjoint
                                 streams and a sharded run reproduces the unsharded one.  0 on a single GPU. */
    int32_t env_pitch;        /* floats between consecutive building rows of the `state` / `out_bldg` planes ([plane][n_bldg][env_pitch], the first
                                 n_env entries of a row used); 0 = n_env.  A batch whose row stride n_env x 4 B is a large power of two (2^20
                                 envs: 4 MiB) makes the 17 x 9 streams of a step alias in the memory system -- 4 - 7 % at 17 x 1 048 576 --
                                 which a pitch of n_env + 256 removes.  Multiple of 4, >= n_env.  Implemented where that regime exists:
                                 CLD_LEAN districts of up to 32 buildings without CLD_KPI / flexible loads / CLD_F64_MAPS (cl_reset_f32,
                                 cl_step_f32, cl_step_observe_f32, cl_observe_f32, cl_rollout_f32, cl_rollout_seq_f32); every other call
                                 returns CL_EINVAL for a pitch other than n_env.  `out_env`, `actions`, `obs` keep their own strides. */
    int32_t reserved0;
} cl_dims;

/* ABI version of the loaded library (== CL_ABI_VERSION of the header it was built from). */
int cl_abi_version(void);

/* Thread-local description of the last error returned on this thread ("" if none). */
const char* cl_last_error(void);

/* Episode start: writes soc[0] / nominal efficiency / nominal capacity into `state`
 * (StorageDevice.reset energy_model.py:797-803, Battery.reset 1237-1242) and zeroes the KPI accumulators
 * (either may be NULL).  Replaces the per-device reset() loop of Building.reset (building.py:2526-2564). */
int cl_reset_f32(const cl_dims* dims, const uint32_t* params, float* state, float* kpi_bldg, float* kpi_env,
                 void* stream);

/* One environment step `t` for every (env, building): apply_actions + update_variables + reward + district
 * sums.  Replaces citylearn.py:1010-1027 for a whole env batch.  `out_bldg`, `out_env` must be non-NULL;
 * `kpi_*` only with CLD_KPI. */
int cl_step_f32(const cl_dims* dims, const uint32_t* params, const float* ts, float* state,
                const float* actions, int64_t act_stride_col, int64_t act_stride_env,
                float* out_bldg, float* out_env, float* kpi_bldg, float* kpi_env, int32_t t, void* stream);

/* Fused rollout: steps t0 .. t0+k_steps-1 in one launch with the per-unit state held in registers.
 * actions == NULL: on-device uniform random policy, a = low + u*(high-low) with
 *   u(seed; env, column, t) = word[t & 3] of Philox4x32-10(counter = (cl_dims.env_offset + env, column, t >> 2, 0), key = seed)  (24 bits)
 * -- the device analogue of Agent.predict's action_space.sample()
 * (agents/base.py:188-209); `act_low/act_high` are [n_act_cols].
 * actions != NULL: open-loop action tensor, element (k, col, env) at
 * actions[k*act_stride_step + col*act_stride_col + env*act_stride_env].
 * `ret_env` [n_env] (optional) accumulates the district reward sum over the k steps (episode return);
 * out_bldg / out_env receive the values of the LAST step.
 * Thermal / outage districts without CLD_WRITE_DETAIL run the pack-generic unit of the thermal step kernels inside the K-step loop (round 6,
 * cl_rollout_full_kernel: one building per wave, two envs per lane on the fp32 map, the Philox blocks of a building's columns cached in LDS for
 * their four steps; `cl_tuning.full_variant` = 1 keeps the scalar-unit kernel for A/B; districts of more than 65 536 action columns keep it too).
 * Districts of more than 32 battery + PV / 16 thermal buildings run building-chunked (round 5): workgroup rows of `cl_tuning.b_chunk` (default 32 /
 * 8) buildings, the last step's chunk partial sums and each chunk's share of the return in the scratch rows of out_bldg's reserved plane
 * (n_chunks x (CL_NQ + 1) rows of n_env floats), folded by ONE cl_finish_kernel launch per call -- out_env / ret_env are final on return.
 * Limits of the fused kernel: no streaming KPIs (CLD_KPI), no flexible loads, no CLD_F64_MAPS (CLD_F64_CHAIN is available), and on a chunked
 * district no reward that couples the buildings inside a step (CLR_MARL: CL_EINVAL) -- cl_rollout_seq_f32 below runs the same K steps as a
 * launch sequence for everything else. */
int cl_rollout_f32(const cl_dims* dims, const uint32_t* params, const float* ts, float* state,
                   const float* actions, int64_t act_stride_step, int64_t act_stride_col, int64_t act_stride_env,
                   const float* act_low, const float* act_high, uint64_t seed,
                   float* out_bldg, float* out_env, float* ret_env, int32_t t0, int32_t k_steps, void* stream);

/* ---- adjacent stage: LSTM indoor-temperature dynamics of LSTMDynamicsBuilding (building.py:3000-3078, dynamics.py) ----
 * lstm_w  [n_bldg][CL_LSTM_NW]            packed LSTM(13->16, 2 layers) + Linear(16->1) weights per building
 * dyn_pre [n_steps][n_bldg][CL_LSTM_NPRE]  host-precomputed env-independent part of the layer-0 gates per (t, building)
 * hist    [CL_LSTM_NHIST][n_bldg][n_env]   rings of the last 12 normalised demand inputs (rows 0-11), indoor temperatures (12-23) and, for a
 *                                          model that takes both demands, second demand inputs (24-35)
 * hidden  [n_bldg][n_env][64]              h0[16], c0[16], h1[16], c1[16] carried across env steps
 * The 64 gate rows of every weight matrix, of the layer-1 bias and of `dyn_pre` are stored pre-multiplied by -log2(e) (gates i, f, o)
 * and -2 log2(e) (gate g): sigmoid(x) = 1 / (1 + 2^z) and tanh(x) = 2 / (1 + 2^z) - 1 then take the accumulated z as it is.
 * (layouts: citylearn_amd/csrc/cl_lstm.h, packer: citylearn_amd/dynamics.py) */
#define CL_LSTM_NW   3360
#define CL_LSTM_NPRE 80
#define CL_LSTM_NHIST 36
#define CL_LSTM_NHIDDEN 64

/* Streaming comfort KPI accumulators `kpi_comfort[CL_NKC][n_bldg][n_env]` (row K of SURVEY 8a for dynamics buildings):
 * running sums / extrema of CostFunction.discomfort and one_minus_thermal_resilience (cost_function.py:224-353),
 * updated by cl_lstm_step_f32 with the temperature it just predicted; finalised on the host (kpi.finalize_comfort). */
#define CL_NKC 10
enum cl_kpi_comfort {
    CLKC_UNMET = 0, CLKC_COLD, CLKC_HOT,                 /* occupied steps outside / below / above the comfort band */
    CLKC_COLD_MIN, CLKC_COLD_MAX, CLKC_COLD_SUM,         /* |min(T - heating set point, 0)| */
    CLKC_HOT_MIN, CLKC_HOT_MAX, CLKC_HOT_SUM,            /* |max(T - cooling set point, 0)| */
    CLKC_UNMET_OUTAGE                                    /* occupied steps outside the band during a power outage */
};

/* Episode start: zero `hist` and `hidden` (LSTMDynamics.reset, dynamics.py:112-127) and, when given, initialise
 * `kpi_comfort` (sums 0, minima +inf, maxima -inf). */
int cl_lstm_reset_f32(const cl_dims* dims, float* hist, float* hidden, float* kpi_comfort, void* stream);

/* After cl_step_f32 of step `t`: push the delivered cooling `cool_dem` [n_bldg][n_env] (out_bldg plane CLO_COOL_DEM)
 * into the window and, once lookback+1 samples exist (t >= 12), run the LSTM over the 12-step window and write the
 * predicted indoor dry-bulb temperature of step t to `indoor_temp` [n_bldg][n_env] (data-file value before that / for
 * buildings without a dynamics model).  Replaces LSTMDynamicsBuilding._update_dynamics_input +
 * update_indoor_dry_bulb_temperature (building.py:3000-3078).
 * `comfort` (optional) receives ComfortReward.calculate per building (reward_function.py:269-334) evaluated on that
 * temperature; `heat_dem` (optional) is the delivered heating plane it compares the cooling demand with;
 * `kpi_comfort` (optional) accumulates the discomfort KPIs.
 * `lstm_wb` (optional, [n_bldg][CL_LSTM_NWB] 16-bit words): the recurrent weight matrices split into three bf16 terms -- or, with
 * CLD_LSTM_F16 in dims->flags, two f16 terms (the first 6144 words of each building's block) -- and laid out as MFMA A-operand
 * fragments (dynamics.pack_lstm_split).  When given, the recurrent products run on the 16-bit matrix cores with split operands
 * (bf16 x 3: dropped terms <= 2^-24 |W||h|; f16 x 2: <= 3 * 2^-22 |W||h| at half the matrix-pipe time; csrc/cl_lstm.h); NULL selects
 * the exact f32-MFMA kernel.
 * A district in which a two-layer model of <= 16 units takes BOTH demands (lstm_w[CLW_DEM2] != 0: delivered heating as a third
 * env-dependent input, its ring in rows 24-35 of `hist`) needs CLD_LSTM_TWO_DEMANDS in dims->flags and `heat_dem`; without the flag that
 * building's indoor_temp is NaN (loud, not wrong), the other buildings are unaffected. */
#define CL_LSTM_NWB 9216
int cl_lstm_step_f32(const cl_dims* dims, const float* lstm_w, const uint16_t* lstm_wb, const float* dyn_pre, const float* cool_dem,
                     const float* heat_dem, float* hist, float* hidden, float* indoor_temp, float* comfort,
                     float* kpi_comfort, int32_t t, void* stream);

/* LSTM shapes the matrix-core kernel does not cover (hidden size up to CL_LSTM_GEN_HMAX, one or two layers; e.g. baeda_3dem's
 * Building_4 = LSTM(11 -> 50, 1 layer)): buildings whose lstm_w[CLW_ACTIVE] is 2 (one layer) or 3 (two layers) are skipped by
 * cl_lstm_step_f32 and advanced by this call (same inputs / outputs, plain fp32 FMAs; call it right after cl_lstm_step_f32).  A model
 * whose inputs hold BOTH cooling_demand and heating_demand (the reference builds the input generically from `input_observation_names`,
 * building.py:3039-3078; lstm_w[CLW_DEM2] != 0, delivered heating as a third env-dependent input) runs on either kernel: on the
 * matrix-core kernel when it is 2 x <= 16 units (CLD_LSTM_TWO_DEMANDS), here otherwise.
 *   gen_w      [n_bldg][gen_w_stride]   WX [H][12] (demand, temperature, second demand input), WHH0 [H][H][4], WIH1 [H][H][4], WHH1 [H][H][4], B1 [H][4], WLIN [H]  (gate order i, f, g, o;
 *                                        H = gen_h, the padded hidden size; csrc/cl_lstm.h, packer dynamics.pack_lstm_generic)
 *   gen_pre    [n_ts_rows][n_bldg][gen_h][4]  env-independent part of the layer-0 gates
 *   gen_hidden [n_bldg][4][gen_h][n_env]      h0, c0, h1, c1 carried across env steps (zero at episode start)
 *   gen_layers = 1 or 2: the deepest model among those buildings (sizes the kernel's LDS: hidden states and, when they fit beside them,
 *                the recurrent matrices) */
#define CL_LSTM_GEN_HMAX 64
int cl_lstm_generic_step_f32(const cl_dims* dims, const float* lstm_w, const float* dyn_pre, const float* gen_w, int64_t gen_w_stride,
                             const float* gen_pre, float* gen_hidden, int32_t gen_h, int32_t gen_layers, const float* cool_dem, const float* heat_dem,
                             float* hist, float* indoor_temp, float* comfort, float* kpi_comfort, int32_t t, void* stream);

/* ---- observation epilogue ----
 * Writes the observation tensor obs[n_env][n_cols] (one contiguous vector per environment, the layout a policy
 * network consumes) for observation row `row` of the episode.  Replaces the per-building dictionary building of
 * Building.observations / CityLearnEnv.observations (building.py:1115-1219, citylearn.py:451-485) and, when the
 * host packs normalised tables, NormalizedObservationWrapper.observation (wrappers.py:131-160).
 *   obs_table [n_rows][n_cols] f32  env-independent value of every column (offset of the affine map for the others)
 *   col_src   [n_cols] i32          -1: env-independent column; else CLOB_SRC(kind, plane, building)
 *   col_scale [n_cols] f32          obs = plane[building][env] * col_scale + obs_table[row][col]
 *   indoor_temp [n_bldg][n_env]     output of cl_lstm_step_f32 (nullable when no column uses CLOB_KIND_TEMP)
 *   obs [n_env][obs_pitch] f32      obs_pitch >= n_cols floats between rows; a multiple of 4 selects the 16-byte store
 *                                   path; pad columns up to the next multiple of 4 (bounded by obs_pitch) are written as 0
 * flags: CLOB_ALL_EXOGENOUS -> every column comes from the table (the observation returned by reset()). */
#define CLOB_KIND_STATE 0           /* plane = enum cl_state */
#define CLOB_KIND_OUT   1           /* plane = enum cl_out */
#define CLOB_KIND_TEMP  2           /* indoor_temp */
#define CLOB_KIND_EXTRA 3           /* caller-provided planes `extra[plane][n_extra_rows][n_env]`, row in the building field
                                       (e.g. cl_flex.flex_out: charging headroom / violation observations) */
#define CLOB_SRC(kind, plane, building) (((kind) << 28) | ((plane) << 20) | (building))
#define CLOB_ALL_EXOGENOUS (1u << 0)
#define CLOB_MAX_DEPS 64
/* One env-dependent column, for the optional HOST-side list `deps` (a compacted copy of col_src / col_scale): when it
 * is given, has n_deps <= CLOB_MAX_DEPS entries and n_cols <= 1024, the list travels in the kernel arguments and the
 * launch takes the fast path (no column-map walk on the device).  Pass deps = NULL, n_deps = -1 otherwise. */
typedef struct cl_obs_dep { int32_t col; int32_t src; float scale; } cl_obs_dep;
int cl_observe_f32(const cl_dims* dims, const float* obs_table, const int32_t* col_src, const float* col_scale,
                   const cl_obs_dep* deps /* host memory, nullable */, int32_t n_deps,
                   const float* state, const float* out_bldg, const float* indoor_temp,
                   const float* extra /* nullable: CLOB_KIND_EXTRA planes */, int32_t n_extra_rows, float* obs, int32_t n_cols,
                   int32_t obs_pitch, int32_t n_rows, int32_t row, uint32_t flags, void* stream);

/* cl_step_f32 followed by cl_observe_f32 of the COMPACT observation form -- every one of the n_cols <= CLOB_MAX_DEPS columns
 * env-dependent and listed in `deps` (host memory), e.g. the [n_env][n_dep] matrix VectorCityLearnEnv(observations='compact') hands
 * out next to the shared row -- for observation row `obs_row` (normally t + 1: Building.observations after next_time_step,
 * citylearn.py:1029-1042).  Same results as the two calls.  Where the step runs as one lean launch at four envs per lane (battery
 * + PV districts of up to 32 buildings from ~50 000 envs up) and the columns are fed by the battery state planes, the net or the
 * reward plane, the wave that stepped a building writes its columns from registers into an LDS tile and the step launch itself
 * streams the tile out: one launch instead of two (17 x 65 536: step + observe 16.8 -> 9.1 us).  Round 6: the thermal / outage step kernels
 * do the same (cl_full.h OBS: districts of up to 32 buildings in one workgroup row, without detail planes / CLD_KPI / CLD_F64_MAPS / a MARL
 * reward; columns fed by the battery and tank state planes, net or reward; 9 x 65 536 with 34 columns: 20.3 -> 12.1 us).  Otherwise it IS the
 * two calls.  A column fed by CLS_B_DEGCAP is refused under CLD_F64_CHAIN (the plane then holds the capacity loss): CL_EINVAL. */
int cl_step_observe_f32(const cl_dims* dims, const uint32_t* params, const float* ts, float* state, const float* actions,
                        int64_t act_stride_col, int64_t act_stride_env, float* out_bldg, float* out_env, float* kpi_bldg, float* kpi_env,
                        int32_t t, const float* obs_table, const int32_t* col_src, const float* col_scale, const cl_obs_dep* deps,
                        int32_t n_deps, float* obs, int32_t n_cols, int32_t obs_pitch, int32_t n_rows, int32_t obs_row, void* stream);

/* ---- flexible loads: EV chargers and washing machines ----
 * Replaces, for a whole env batch, Charger.update_connected_electric_vehicle_soc (electric_vehicle_charger.py:297-334),
 * WashingMachine.start_cycle / next_time_step (energy_model.py:1289-1330), the chargers / washing-machine terms of
 * Building.update_variables (building.py:2657-2693), CityLearnEnv.simulate_unconnected_ev_soc and
 * associate_chargers_to_electric_vehicles (citylearn.py:1353-1474) and Electric_Vehicles_Reward_Function
 * (reward_function.py:389-531).  Everything that depends only on the charger schedules is folded into the tables by
 * the host packer (citylearn_amd/flex.py); the device keeps the per-(EV, env) battery state.
 *
 *   ev_params      [n_ev][CL_NP]                 battery block of an EV: the CLP_L_* words of a `params` row
 *   ev_ts          [n_rows][n_ev][CL_NEVF]       begin-of-step SoC rule of (row, EV)            (cl_ev_feat)
 *   charger_params [n_flex_bldg][CL_MAXC][CL_NCP]           (cl_charger_param) one slot per charger of the building
 *   charger_ts     [n_rows][n_flex_bldg][CL_MAXC][CL_NCF]   (cl_charger_feat)
 *   wm_params      [n_flex_bldg][CL_MAXW][CL_NWP]           word 0: action column (i32, -1 = inactive)
 *   wm_ts          [n_rows][n_flex_bldg][CL_MAXW][CL_NWF]   (cl_wm_feat)
 * (slot tables: every address a wave needs depends only on its building and row, so the table reads of a step are ONE
 *  round of independent scalar loads -- a first/count indirection per building cost two more dependent round trips)
 *   ev_state       [CL_NEVS][n_ev][n_env]        soc written last, Battery.efficiency, degraded capacity (same meaning as CLS_B_*)
 *   wm_state       [n_flex_bldg * CL_MAXW][n_env]  WashingMachine.initiated (0 / 1), by slot
 *   flex_out       [CL_NX][n_flex_bldg][n_env]   per-building results consumed by the step kernel (cl_flex_out)
 *   charger_out    [2][n_flex_bldg * CL_MAXC][n_env]  optional detail by slot: charger electricity_consumption[t], past_charging_action_values_kwh[t]
 *   drift          [n_rows][n_ev]                optional: multipliers of the unconnected-EV SoC drift (citylearn.py:1468-1472) to
 *                                                replay; NULL draws N(1, 0.2) per (env, EV, t) from Philox4x32-10 keyed by `seed`
 * Rows: step t of env block g reads row t + env_row0[g] (row t without offsets), exactly like `ts`. */
#define CL_NEVF 4
enum cl_ev_feat {
    CLEV_RULE_STEP = 0,   /* how soc[row] starts when the step is entered from row - 1:  v >= 0: v,  CLEV_ZERO, CLEV_DRIFT */
    CLEV_RULE_LAST,       /* the same on the last step of an episode (no simulate_unconnected_ev_soc there, citylearn.py:1419-1420) */
    CLEV_RULE_RESET,      /* ... when the episode starts on this row:  v >= 0: v,  CLEV_KEEP: battery initial_soc */
    CLEV_CONNECTED        /* 1 when a charger holds this EV on this row (its charger advances it), else 0 */
};
#define CLEV_ZERO  (-1.0f)
#define CLEV_DRIFT (-2.0f)
#define CLEV_KEEP  (-3.0f)
#define CL_NCP 48
#define CL_CURVE_MAX 8      /* points per charger efficiency curve */
enum cl_charger_param {
    CLC_ACT_COL = 0,      /* i32 action column, -1 = inactive */
    CLC_MAX_CHARGE, CLC_MIN_CHARGE, CLC_MAX_DISCHARGE, CLC_MIN_DISCHARGE,   /* kW */
    CLC_EFF, CLC_INV_EFF, /* Charger.efficiency and its reciprocal */
    CLC_DT_HOURS,         /* seconds_per_time_step / 3600 */
    /* optional charge / discharge efficiency curves over |action| (Charger.get_efficiency = np.interp,
       electric_vehicle_charger.py:264-295): point count (i32, 0 = use CLC_EFF), then CL_CURVE_MAX x, CL_CURVE_MAX y */
    CLC_CURVE_CHARGE_N = 8, CLC_CURVE_CHARGE_X = 9, CLC_CURVE_CHARGE_Y = 17,
    CLC_CURVE_DISCHARGE_N = 25, CLC_CURVE_DISCHARGE_X = 26, CLC_CURVE_DISCHARGE_Y = 34
};
#define CL_MAXC 4           /* charger slots per building */
#define CL_MAXW 2           /* washing-machine slots per building */
#define CL_NCF 8
enum cl_charger_feat {
    CLCT_EV = 0,          /* index of the connected EV (state 1 and a known id) as a float, -1 = none, CLCT_EMPTY = no charger in this slot */
    CLCT_REQUIRED_SOC,    /* electric_vehicle_required_soc_departure */
    CLCT_DEPARTURE,       /* electric_vehicle_departure_time [steps] */
    CLCT_RULE_STEP,       /* copies of the connected EV's CLEV_RULE_STEP / CLEV_RULE_LAST on this row */
    CLCT_RULE_LAST
};
#define CLCT_EMPTY (-2.0f)
#define CL_NWP 2
#define CL_NWF 4
enum cl_wm_feat {
    CLWT_OPEN = 0,        /* 1 when start/end are set and start <= step <= end on this row; CLWT_EMPTY = no washing machine in this slot */
    CLWT_NEW_WINDOW,      /* 1 when (start, end) differ from the previous row: clears `initiated` (energy_model.py:1303-1312) */
    CLWT_LOAD             /* what start_cycle books on this row: the load profile summed over the offsets still inside the episode */
};
#define CLWT_EMPTY (-1.0f)
#define CL_NEVS 3
#define CL_MAXPH 4          /* charging-constraint phases per building */
#define CL_NX 11
enum cl_flex_out {
    CLX_LOAD = 0,         /* chargers + washing machines electricity [kWh], added to the building's net */
    CLX_CHARGERS,         /* chargers only (removed again for evaluate()'s baseline, building.py:345-366) */
    CLX_RW_K0, CLX_RW_KNEG, CLX_RW_KPOS,  /* Electric_Vehicles_Reward_Function: reward_b = (K0 + [net<0] KNEG + [net>0] KPOS) / (1 + |MARL_b|) - penalty */
    /* buildings with charging constraints only (building.py:901-989): */
    CLX_VIOLATION,        /* charging_constraint_violation_kwh of this step */
    CLX_HEADROOM,         /* charging_building_headroom_kw: building limit - admitted charging power */
    CLX_HEADROOM_PHASE0   /* .. + CL_MAXPH - 1: charging_phase_<name>_headroom_kw */
};
enum cl_ev_weight { CLEW_BATTERY_LIMITS = 0, CLEW_SOC_IMPOSSIBLE, CLEW_SOC_UNDER, CLEW_CLOSE_SOC, CLEW_SELF_EV_CONSUMPTION,
                    CLEW_EXTRA_SELF_PRODUCTION, CLEW_PENALTY_COEFFICIENT /* charging_constraint_penalty_coefficient */, CL_NEW };
/* charging constraints of a building (`cons_params[n_flex_bldg][CL_NCC]`, nullable = no building has any): the positive
 * charger requests action * max_charging_power are scaled down so that their sum stays below the building limit and each
 * phase's sum below the phase limit; the excess is the violation (Building._apply_charging_constraints_to_actions). */
#define CL_NCC 12
enum cl_cons_param {
    CLCC_FLAGS = 0,       /* u32: bit 0 = this building has constraints */
    CLCC_BUILDING_LIMIT,  /* kW, < 0 = none */
    CLCC_PHASE_LIMIT0,    /* .. + CL_MAXPH - 1: kW, < 0 = none / unused */
    CLCC_PHASE_MASK0 = CLCC_PHASE_LIMIT0 + CL_MAXPH   /* .. + CL_MAXPH - 1: u32 bit j = charger slot j is on this phase */
};
typedef struct cl_flex {
    int32_t n_ev, n_flex_bldg, n_rows, reserved;
    const uint32_t* ev_params;
    const float* ev_ts;
    const uint32_t* charger_params;
    const float* charger_ts;
    const uint32_t* wm_params;
    const float* wm_ts;
    const uint32_t* cons_params;  /* nullable */
    float* ev_state;
    float* wm_state;
    float* flex_out;
    float* charger_out;       /* nullable */
    const float* drift;       /* nullable */
    uint64_t seed;
    float weights[8];         /* cl_ev_weight */
} cl_flex;

/* Episode start: EV SoC = CLEV_RULE_RESET of each env block's first row, nominal efficiency / capacity, washing
 * machines idle (ElectricVehicle.reset, Charger.reset, WashingMachine.reset + the reset-time association, citylearn.py:1871-1874). */
int cl_flex_reset_f32(const cl_dims* dims, const cl_flex* flex, void* stream);

/* cl_step_f32 for a district with flexible loads: advances chargers / EVs / washing machines of step `t` (one extra
 * launch), then runs the step with their load added to the flagged buildings' nets.  flex == NULL is cl_step_f32. */
int cl_step_flex_f32(const cl_dims* dims, const uint32_t* params, const float* ts, float* state,
                     const float* actions, int64_t act_stride_col, int64_t act_stride_env,
                     float* out_bldg, float* out_env, float* kpi_bldg, float* kpi_env, const cl_flex* flex,
                     int32_t t, void* stream);

/* The same K steps as cl_rollout_f32 for everything the fused kernel does not hold in registers: districts with flexible loads
 * (`flex`: charger / EV / washing-machine state lives in HBM between steps), streaming KPIs (CLD_KPI: `kpi_bldg` / `kpi_env`
 * updated after every step, exactly as K calls of cl_step_f32 would), districts of any size (building-chunked launches).  The K
 * steps are K x (policy, [flex], step, [kpi], return) launches enqueued on `stream` (capturable in a hipGraph).  Same action
 * sources as cl_rollout_f32: open-loop `actions` [k_steps][n_act_cols][n_env] (strides in floats) or, with actions == NULL, the
 * on-device policy a = low + u (high - low), u = cl_philox_uniform(seed, env, column, t), generated four steps at a time into the
 * scratch planes `policy_actions` [4][n_act_cols][n_env] (required then; n_env a multiple of 4).
 * `flex`, `kpi_bldg`, `kpi_env` are nullable; `ret_env` [n_env] (optional) accumulates the district reward; out_bldg / out_env
 * hold the LAST step's values. */
int cl_rollout_seq_f32(const cl_dims* dims, const uint32_t* params, const float* ts, float* state,
                       const float* actions, int64_t act_stride_step, int64_t act_stride_col, int64_t act_stride_env,
                       const float* act_low, const float* act_high, uint64_t seed, float* policy_actions,
                       float* out_bldg, float* out_env, float* ret_env, float* kpi_bldg, float* kpi_env, const cl_flex* flex,
                       int32_t t0, int32_t k_steps, void* stream);

/* Deferred finish (cl_tuning.finish = 3): bring `out_env` up to date with step `t` -- the last step enqueued on `stream` -- by folding
 * the chunk partial sums that step left in `out_bldg`'s reserved plane (reference: the district sums of CityLearnEnv.update_variables,
 * citylearn.py:1888-1918, which the reference forms inside every step).  Idempotent; a no-op (one tiny launch that finds no marker, or
 * none at all for districts of up to 32 buildings) when step `t` finished its own sums.  Call it before reading `out_env` and at the end
 * of a captured step sequence; between two deferred steps it is not needed (each launch folds its predecessor's sums). */
int cl_finish_f32(const cl_dims* dims, float* out_bldg, float* out_env, int32_t t, void* stream);

/* Philox4x32-10 reference draw used by cl_rollout_f32 (host-callable so tests can reproduce the policy):
 * returns u in [0,1) for (seed, env, col, t). */
float cl_philox_uniform(uint64_t seed, uint32_t env, uint32_t col, uint32_t t);

#ifdef __cplusplus
}
#endif
#endif /* CITYLEARN_AMD_H */
