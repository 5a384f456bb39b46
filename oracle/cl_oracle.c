/*
 * cl_oracle.c -- plain-C restatement of the CityLearn step arithmetic (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Same algorithm as oracle/oracle.py (which is pinned bit-exactly to the reference), written for speed so that
 * (a) parity tests can run thousands of envs, and (b) bench.py has a CPU baseline ("port") to time on the GPU
 * box's host cores.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * Parity pinning: tests/test_oracle_golden.py checks this file against the committed reference trajectories
 * (tests/golden/<fixture>/reference.npz) -- max |error| <= 1e-6 on every quantity.
 *
 * Arithmetic is double with float rounding wherever the reference stores into a float32 series
 * (energy_model.py:151-155, 797-803; building.py:2555-2564), which is what makes it track the reference.
 * Reference citations are relative to /root/reference/citylearn/.
 *
 * Build: gcc -O2 -fPIC -shared -fopenmp oracle/cl_oracle.c -o oracle/libcl_oracle.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define ZDP 1e-6

/* per-building parameters, doubles */
enum {
    OP_FLAGS = 0,        /* bit0 battery present, bit1 heating is heat pump, bit2 dhw is heat pump, bit3 simulate outage,
                            bit4 dynamics building */
    OP_DT, OP_R,
    OP_B_CAP, OP_B_POW, OP_B_LOSS, OP_B_CLC, OP_B_DOD, OP_B_EFF0, OP_B_SOC0,
    OP_B_CPC_X /*3*/, OP_B_CPC_Y = OP_B_CPC_X + 3 /*3*/, OP_B_PEC_X = OP_B_CPC_Y + 3 /*5*/, OP_B_PEC_Y = OP_B_PEC_X + 5 /*5*/,
    OP_CS = OP_B_PEC_Y + 5, /* cap, loss, eff, soc0, maxin, maxout */
    OP_HS = OP_CS + 6, OP_DS = OP_HS + 6,
    OP_CD_POW = OP_DS + 6, OP_CD_EFF, OP_CD_TC,
    OP_HD_POW, OP_HD_EFF, OP_HD_TH,
    OP_DD_POW, OP_DD_EFF, OP_DD_TH,
    OP_DYN_WARMUP,
    OP_ACT_CS, OP_ACT_HS, OP_ACT_DS, OP_ACT_ES, OP_ACT_CD, OP_ACT_HD, OP_ACT_COH,   /* action columns, -1 inactive */
    OP_N
};
/* per-(t, building) series row, doubles */
enum { OT_NSL = 0, OT_SOLAR_WKW, OT_COOL, OT_HEAT, OT_DHW, OT_TOUT, OT_PRICE, OT_CARBON, OT_OUTAGE, OT_HVAC, OT_PV_POW, OT_N };
/* state per unit, floats: soc, eff(double kept separately), ... */
enum { OS_SOC = 0, OS_EFF, OS_DEGCAP, OS_CS, OS_HS, OS_DS, OS_N };
enum { OO_NET = 0, OO_REWARD, OO_EB, OO_COOL_DEM, OO_C_COOL, OO_C_HEAT, OO_C_DHW, OO_C_NS, OO_COST, OO_EMISSION, OO_N };

int cl_oracle_np(void) { return OP_N; }
int cl_oracle_nt(void) { return OT_N; }
int cl_oracle_ns(void) { return OS_N; }
int cl_oracle_no(void) { return OO_N; }

static inline double f32(double x) { return (double)(float)x; }

/* HeatPump.get_cop, float32 arithmetic like the reference (energy_model.py:216-250) */
static double cop_hp(double eff, double target, double t_out, int heating) {
    float num = (float)(eff * (target + 273.15));
    float den = heating ? (float)target - (float)t_out : (float)t_out - (float)target;
    float c = num / den;
    if (c < 0 || c > 20 || c != c) c = 20;   /* NaN cannot occur for finite inputs; inf -> 20 */
    return c;
}

static int interp_index(double x, const double* xs, int n) {
    /* max(0, argmax(x <= xs) - 1): first i with x <= xs[i]; all-false -> argmax = 0 (energy_model.py:1083) */
    int first = 0, found = 0;
    for (int i = 0; i < n; ++i)
        if (x <= xs[i]) { first = i; found = 1; break; }
    if (!found) first = 0;
    return first - 1 < 0 ? 0 : first - 1;
}

typedef struct {
    double prev_soc, soc, eb;      /* float32-valued */
} tank_t;

/* StorageDevice.charge (energy_model.py:719-768) */
static void base_charge(tank_t* k, double energy, double cap, double loss_r, double rte, double r) {
    energy *= r;
    double e_init = f32(k->prev_soc * cap) * (1.0 - loss_r);     /* float32 product, then float64 (r is np.float64) */
    if (e_init < 0) e_init = 0;
    double e_fin;
    if (energy >= 0) { e_fin = e_init + energy * rte; if (e_fin > cap) e_fin = cap; }
    else { e_fin = e_init + energy / rte; if (e_fin < 0) e_fin = 0; }
    k->soc = f32(e_fin / (cap > ZDP ? cap : ZDP));
    double d = e_fin - e_init;
    k->eb = f32(d >= 0 ? d / rte : d * rte);
}

/* StorageTank.charge (energy_model.py:850-870) */
static void tank_charge(tank_t* k, double energy, const double* tp, double r) {
    energy *= r;
    if (energy >= 0) { if (energy > tp[4]) energy = tp[4]; }
    else { if (energy < -tp[5]) energy = -tp[5]; }
    base_charge(k, energy, tp[0], tp[1] * r, sqrt(tp[2]), r);
}

typedef struct {
    const double* p;
    const double* row;
    double r, dt;
    int outage;
    double c_cool, c_heat, c_dhw, c_ns, c_b;   /* float32-valued accumulators */
    double cop_c, cop_h, cop_d;
} unit_t;

static double flex(const unit_t* u) {            /* building.py:640-668 */
    if (!u->outage) return INFINITY;
    double solar = -(u->row[OT_PV_POW] * u->row[OT_SOLAR_WKW] / 1000.0);
    double used = f32(f32(f32(f32(f32(u->c_cool * u->r) + f32(u->c_heat * u->r)) + f32(u->c_dhw * u->r)) + f32(u->c_ns * u->r)) + f32(u->c_b * u->r));
    double cap = fabs(solar) - used;
    return cap > 0 ? cap : 0;
}

static double max_out(const unit_t* u, double pow, double c, double cop) {
    double avail = pow - c * u->r;
    double f = flex(u);
    return (f < avail ? f : avail) * cop;
}

static void end_use_device(unit_t* u, double* c, double demand, const tank_t* k, double pow, double cop, double* e_dev) {
    double storage_output = k->eb < 0 ? -k->eb : 0.0;           /* building.py:525-541 */
    double mo = max_out(u, pow, *c, cop);
    double out = demand - storage_output;
    if (mo < out) out = mo;
    *e_dev = f32(out);
    double cons = out / cop;
    *c = f32(*c + (cons > 0 ? cons : 0));
}

static void end_use_storage(unit_t* u, double* c, double demand, tank_t* k, const double* tp, double action, double cscale,
                            double pow, double cop) {
    double energy = action * cscale;
    if (energy > 0.0) { double mo = max_out(u, pow, *c, cop); if (mo < energy) energy = mo; }
    else { if (energy < -demand) energy = -demand; }
    tank_charge(k, energy / u->r, tp, u->r);
    double charged = k->eb > 0 ? k->eb : 0.0;
    *c = f32(*c + charged / cop);
}

/* update_electrical_storage + Battery.charge (building.py:1791-1812, energy_model.py:1027-1141) */
static void battery_charge(unit_t* u, tank_t* es, double* eff, double* degcap, double a_es) {
    const double* p = u->p;
    if (!((int)p[OP_FLAGS] & 1)) return;
    double energy = a_es * p[OP_B_POW] * u->dt;
    double f = flex(u);
    if (f < energy) energy = f;
    energy = energy / u->r * u->r;                       /* _convert_energy_for_storage, then charge()'s own * r */
    double action_energy = energy;
    const double cap = p[OP_B_CAP], powr = p[OP_B_POW];
    double e_init = f32(es->prev_soc * cap) * (1.0 - p[OP_B_LOSS] * u->r);
    if (e_init < 0) e_init = 0;
    double socn = e_init / (cap > ZDP ? cap : ZDP);
    const double* cx = p + OP_B_CPC_X; const double* cy = p + OP_B_CPC_Y;
    int i = interp_index(socn, cx, 3);
    double pmax = powr * (cy[i] + (cy[i + 1] - cy[i]) * (socn - cx[i]) / (cx[i + 1] - cx[i]));
    double x;
    if (energy >= 0) {
        double avail = powr - u->c_b * u->r, wrt = *degcap - e_init;
        double m = pmax; if (avail < m) m = avail; if (wrt < m) m = wrt; if (energy < m) m = energy;
        energy = m;
        x = action_energy < pmax ? action_energy : pmax;
    } else {
        double lim = f32(f32(es->prev_soc - (1.0 - p[OP_B_DOD])) * cap) * sqrt(*eff);
        lim = lim > 0 ? -lim : -0.0;
        double m = -pmax; if (lim > m) m = lim; if (energy > m) m = energy;
        energy = m;
        x = fabs(action_energy) < pmax ? fabs(action_energy) : pmax;
    }
    x = fabs(x) / (powr > ZDP ? powr : ZDP);
    const double* ex = p + OP_B_PEC_X; const double* ey = p + OP_B_PEC_Y;
    i = interp_index(x, ex, 5);
    *eff = ey[i] + (x - ex[i]) * (ey[i + 1] - ey[i]) / (ex[i + 1] - ex[i]);
    base_charge(es, energy, cap, p[OP_B_LOSS] * u->r, sqrt(*eff), u->r);
    double deg = f32(p[OP_B_CLC] * cap * fabs(es->eb) / (2 * (*degcap > ZDP ? *degcap : ZDP))) * u->r;
    *degcap = *degcap - deg; if (*degcap < 0) *degcap = 0;
    u->c_b = f32(u->c_b + es->eb);
}

/*
 * One step `t` for n_env x n_bldg units (OpenMP over envs).
 *   params [B][OP_N] doubles; ts [T][B][OT_N] doubles; state [E][B][OS_N] doubles (float32-valued except eff);
 *   actions [A][E] floats; out [E][B][OO_N] doubles; out_env [E][4] doubles (net, cost, emission, reward sum)
 *   reward_kind: 0 default, 1 MARL, 2 IndependentSAC, 3 SolarPenalty
 */
void cl_oracle_step(int n_env, int n_bldg, const double* params, const double* ts, double* state, const float* actions,
                    double* out, double* out_env, int t, int t0_quirk, int reward_kind, double exponent) {
#pragma omp parallel for schedule(static)
    for (int e = 0; e < n_env; ++e) {
        double d_net = 0, d_cost = 0, d_em = 0;
        for (int b = 0; b < n_bldg; ++b) {
            const double* p = params + (size_t)b * OP_N;
            const double* row = ts + ((size_t)t * n_bldg + b) * OT_N;
            double* S = state + ((size_t)e * n_bldg + b) * OS_N;
            double* O = out + ((size_t)e * n_bldg + b) * OO_N;
            const int flags = (int)p[OP_FLAGS];
            unit_t u;
            u.p = p; u.row = row; u.r = p[OP_R]; u.dt = p[OP_DT];
            u.outage = (flags & 8) && row[OT_OUTAGE] != 0.0;
            u.cop_c = cop_hp(p[OP_CD_EFF], p[OP_CD_TC], row[OT_TOUT], 0);
            u.cop_h = (flags & 2) ? cop_hp(p[OP_HD_EFF], p[OP_HD_TH], row[OT_TOUT], 1) : p[OP_HD_EFF];
            u.cop_d = (flags & 4) ? cop_hp(p[OP_DD_EFF], p[OP_DD_TH], row[OT_TOUT], 1) : p[OP_DD_EFF];
            const double solar = -(row[OT_PV_POW] * row[OT_SOLAR_WKW] / 1000.0);
            const double t0_heat_div = (flags & 2) ? u.cop_h : p[OP_DD_EFF];        /* building.py:2626-2634 */
            const int first = t0_quirk && t == 0;
            u.c_cool = u.c_heat = u.c_dhw = u.c_ns = u.c_b = 0;
            if (first) {   /* reset-time update_variables (citylearn.py:1884 -> building.py:2618-2652) */
                u.c_cool = f32(row[OT_COOL] / u.cop_c); u.c_heat = f32(row[OT_HEAT] / t0_heat_div);
                u.c_dhw = f32(row[OT_DHW] / u.cop_d); u.c_ns = f32(row[OT_NSL]);
            }
#define ACT(slot) ((int)p[slot] >= 0 ? (double)actions[(size_t)(int)p[slot] * n_env + e] : 0.0)
            double a_cs = ACT(OP_ACT_CS), a_hs = ACT(OP_ACT_HS), a_ds = ACT(OP_ACT_DS), a_es = ACT(OP_ACT_ES);
            double a_cd = ACT(OP_ACT_CD), a_hd = ACT(OP_ACT_HD);
            if ((int)p[OP_ACT_COH] >= 0) { double a = ACT(OP_ACT_COH); a_cd = fabs(a < 0 ? a : 0); a_hd = fabs(a > 0 ? a : 0); }
#undef ACT
            /* partial-load demand (building.py:3080-3158) */
            double cool_dem = row[OT_COOL], heat_dem = row[OT_HEAT], dhw_dem = row[OT_DHW];
            if ((flags & 16) && t >= (int)p[OP_DYN_WARMUP]) {
                int coh = (int)p[OP_ACT_COH] >= 0;
                int hv = (int)row[OT_HVAC];
                if ((int)p[OP_ACT_CD] >= 0 || coh) {
                    if (hv == 1 || hv == 3) {
                        double power = a_cd * p[OP_CD_POW] * u.dt, avail = p[OP_CD_POW] - u.c_cool * u.r;
                        cool_dem = f32((power < avail ? power : avail) * u.cop_c);
                    } else cool_dem = 0;
                }
                if ((int)p[OP_ACT_HD] >= 0 || coh) {
                    if (hv == 2 || hv == 3) {
                        double power = a_hd * p[OP_HD_POW], avail = p[OP_HD_POW] - u.c_heat * u.r;
                        heat_dem = f32((power < avail ? power : avail) * u.cop_h);
                    } else heat_dem = 0;
                }
            }
            tank_t cs = {S[OS_CS], 0, 0}, hs = {S[OS_HS], 0, 0}, ds = {S[OS_DS], 0, 0};
            tank_t es = {S[OS_SOC], 0, 0};
            double eff = S[OS_EFF], degcap = S[OS_DEGCAP];
            double e_cool = f32(cool_dem), e_heat = f32(heat_dem), e_dhw = f32(dhw_dem), e_ns = f32(row[OT_NSL]);
            /* order (building.py:1567-1634): battery first when discharging, storage before device when discharging */
            if (a_es < 0.0) battery_charge(&u, &es, &eff, &degcap, a_es);
            {
                const double* tcs = p + OP_CS; const double* ths = p + OP_HS; const double* tds = p + OP_DS;
                if (a_cs < 0.0) {
                    end_use_storage(&u, &u.c_cool, cool_dem, &cs, tcs, a_cs, tcs[0], p[OP_CD_POW], u.cop_c);
                    end_use_device(&u, &u.c_cool, cool_dem, &cs, p[OP_CD_POW], u.cop_c, &e_cool);
                } else {
                    end_use_device(&u, &u.c_cool, cool_dem, &cs, p[OP_CD_POW], u.cop_c, &e_cool);
                    end_use_storage(&u, &u.c_cool, cool_dem, &cs, tcs, a_cs, tcs[0], p[OP_CD_POW], u.cop_c);
                }
                if (a_hs < 0.0) {
                    end_use_storage(&u, &u.c_heat, heat_dem, &hs, ths, a_hs, tcs[0] * u.dt, p[OP_HD_POW], u.cop_h);
                    end_use_device(&u, &u.c_heat, heat_dem, &hs, p[OP_HD_POW], u.cop_h, &e_heat);
                } else {
                    end_use_device(&u, &u.c_heat, heat_dem, &hs, p[OP_HD_POW], u.cop_h, &e_heat);
                    end_use_storage(&u, &u.c_heat, heat_dem, &hs, ths, a_hs, tcs[0] * u.dt, p[OP_HD_POW], u.cop_h);
                }
                if (a_ds < 0.0) {
                    end_use_storage(&u, &u.c_dhw, dhw_dem, &ds, tds, a_ds, ths[0] * u.dt, p[OP_DD_POW], u.cop_d);
                    end_use_device(&u, &u.c_dhw, dhw_dem, &ds, p[OP_DD_POW], u.cop_d, &e_dhw);
                } else {
                    end_use_device(&u, &u.c_dhw, dhw_dem, &ds, p[OP_DD_POW], u.cop_d, &e_dhw);
                    end_use_storage(&u, &u.c_dhw, dhw_dem, &ds, tds, a_ds, ths[0] * u.dt, p[OP_DD_POW], u.cop_d);
                }
                double f = flex(&u);
                double d = row[OT_NSL] < f ? row[OT_NSL] : f;      /* building.py:1784-1789 */
                e_ns = f32(d);
                u.c_ns = f32(u.c_ns + d);
            }
            if (!(a_es < 0.0)) battery_charge(&u, &es, &eff, &degcap, a_es);
            if (first) {   /* second t == 0 pass of update_variables (building.py:2618-2652) */
                u.c_cool = f32(u.c_cool + (e_cool + cs.eb) / u.cop_c);
                u.c_heat = f32(u.c_heat + (e_heat + hs.eb) / t0_heat_div);
                u.c_dhw = f32(u.c_dhw + (e_dhw + ds.eb) / u.cop_d);
                u.c_ns = f32(u.c_ns + e_ns);
                u.c_b = f32(u.c_b + es.eb);
            }
            double net = 0.0;
            if (!u.outage)
                net = f32(f32(f32(f32(f32(u.c_cool * u.r) + f32(u.c_heat * u.r)) + f32(u.c_dhw * u.r)) + f32(u.c_ns * u.r)) + f32(u.c_b * u.r)) + solar;
            double net32 = f32(net);
            S[OS_SOC] = es.soc; S[OS_EFF] = eff; S[OS_DEGCAP] = degcap; S[OS_CS] = cs.soc; S[OS_HS] = hs.soc; S[OS_DS] = ds.soc;
            O[OO_NET] = net32; O[OO_EB] = es.eb;
            O[OO_COOL_DEM] = e_cool + fabs(cs.eb < 0 ? cs.eb : 0.0);
            /* Device.electricity_consumption = accumulator * time_step_ratio (energy_model.py:118) */
            O[OO_C_COOL] = u.c_cool * u.r; O[OO_C_HEAT] = u.c_heat * u.r; O[OO_C_DHW] = u.c_dhw * u.r; O[OO_C_NS] = u.c_ns * u.r;
            O[OO_COST] = f32(net * row[OT_PRICE]);
            double em = net * row[OT_CARBON];
            O[OO_EMISSION] = f32(em > 0 ? em : 0);
            d_net = f32(d_net + net32); d_cost = f32(d_cost + O[OO_COST]); d_em = f32(d_em + O[OO_EMISSION]);
            double rw;
            switch (reward_kind) {
            case 2: rw = -net32 < 0 ? -net32 : 0; break;
            case 3: {
                double sg = net32 > 0 ? 1 : (net32 < 0 ? -1 : 0), an = fabs(net32);
                rw = 0;
                if (p[OP_CS] > ZDP) rw += -(1.0 + sg * cs.soc) * an;
                if (p[OP_HS] > ZDP) rw += -(1.0 + sg * hs.soc) * an;
                if (p[OP_DS] > ZDP) rw += -(1.0 + sg * ds.soc) * an;
                if (p[OP_B_CAP] > ZDP) rw += -(1.0 + sg * es.soc) * an;
                break;
            }
            case 1: rw = net32; break;     /* finished below */
            default: { double m = net32 > 0 ? net32 : 0; rw = -(exponent == 1.0 ? m : pow(m, exponent)); }
            }
            O[OO_REWARD] = rw;
        }
        double d_rw = 0;
        for (int b = 0; b < n_bldg; ++b) {
            double* O = out + ((size_t)e * n_bldg + b) * OO_N;
            if (reward_kind == 1) {
                double n = -O[OO_NET];
                double sg = n > 0 ? 1 : (n < 0 ? -1 : 0);
                O[OO_REWARD] = sg * 0.01 * n * n * (d_net > 0 ? d_net : 0);
            }
            d_rw += O[OO_REWARD];
        }
        out_env[(size_t)e * 4 + 0] = d_net; out_env[(size_t)e * 4 + 1] = d_cost; out_env[(size_t)e * 4 + 2] = d_em;
        out_env[(size_t)e * 4 + 3] = d_rw;
    }
}
