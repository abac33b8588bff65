/* stepper.c — ray ODE integrator.  ORACLE (test infrastructure).
 *
 * Restates `Environment::cast_ray_stepper(h0, ang, straight)` + `PathStepper::set_step_size` +
 * `Iterator::next -> RayState{x,h,dh}` of crate `atm-refraction` 0.6 (source absent, PARITY
 * UNPINNED).  Call sites in the reference: utils.rs:142-145,160; rectilinear.rs:134-137,181;
 * ray_path.rs:71-77.  The first next() returns the state AFTER one step (utils.rs:147-160 pushes
 * the start point itself).  Physics (derived):
 *   flat   (Snell, n cos(theta) = const):  h'' = (1 + h'^2) n'(h)/n(h),      h'(0) = tan(ang)
 *   sphere (Bouguer, n r cos(theta) = const), r(phi), x = R phi:
 *          r'' = r + 2 r'^2 / r + (r^2 + r'^2) n'(h)/n(h),                   r'(0) = r0 tan(ang)
 *   straight: flat h = h0 + x tan(ang);  sphere r = r0 cos(ang)/cos(ang + x/R).
 * Integrator: classical RK4 with step `step` in x (d phi = step / R); x accumulates by `x += step`.
 * Evaluation order (round 4; the crate's is unknown, see atmosphere.c): stage points and the final combination are fused
 * multiply-adds, and the spherical right-hand side is taken over ONE common denominator,
 *          r'' = r + (2 r'^2 n + (r^2 + r'^2) r n') / (r n).
 */
#include "oracle.h"
#include "oracle_math.h"

void oracle_stepper_init(oracle_stepper* s, const oracle_env_atm* atm, int spherical, double radius, double h0,
                         double ang_rad, int straight, double step) {
  s->atm = atm;
  s->spherical = spherical;
  s->straight = straight;
  s->radius = radius;
  s->step = step;
  s->x = 0.0;
  s->h0 = h0;
  s->ang = ang_rad;
  if (spherical) {
    s->a = h0 + radius;
    s->b = s->a * om_tan(ang_rad);
  } else {
    s->a = h0;
    s->b = om_tan(ang_rad);
  }
}

static double accel_flat(const oracle_env_atm* atm, double h, double v) {
  double n = oracle_n(atm, h);
  double dn = oracle_dn(atm, h);
  return om_fma(v, v, 1.0) * dn / n;
}

static double accel_sph(const oracle_env_atm* atm, double radius, double r, double v) {
  double h = r - radius;
  double n = oracle_n(atm, h);
  double dn = oracle_dn(atm, h);
  double v2 = v * v;
  double s = om_fma(r, r, v2);
  double num = om_fma(v2 + v2, n, s * r * dn);
  return r + num / (r * n);
}

oracle_ray_state oracle_stepper_next(oracle_stepper* s) {
  oracle_ray_state out;
  if (s->straight) {
    s->x = s->x + s->step;
    if (s->spherical) {
      double r0 = s->h0 + s->radius;
      double phi = s->x / s->radius;
      double c = om_cos(s->ang + phi);
      out.h = r0 * om_cos(s->ang) / c - s->radius;
      out.dh = om_tan(s->ang + phi);
    } else {
      out.h = s->h0 + s->x * om_tan(s->ang);
      out.dh = om_tan(s->ang);
    }
    out.x = s->x;
    return out;
  }
  {
    double d = s->spherical ? s->step / s->radius : s->step;
    double half = 0.5 * d, sixth = d / 6.0;
    double a = s->a, b = s->b;
    double k1a, k1b, k2a, k2b, k3a, k3b, k4a, k4b;
#define ACC(pa, pb) (s->spherical ? accel_sph(s->atm, s->radius, (pa), (pb)) : accel_flat(s->atm, (pa), (pb)))
    k1a = b;
    k1b = ACC(a, b);
    k2a = om_fma(half, k1b, b);
    k2b = ACC(om_fma(half, k1a, a), k2a);
    k3a = om_fma(half, k2b, b);
    k3b = ACC(om_fma(half, k2a, a), k3a);
    k4a = om_fma(d, k3b, b);
    k4b = ACC(om_fma(d, k3a, a), k4a);
#undef ACC
    s->a = om_fma(sixth, om_fma(2.0, k3a, om_fma(2.0, k2a, k1a)) + k4a, a);
    s->b = om_fma(sixth, om_fma(2.0, k3b, om_fma(2.0, k2b, k1b)) + k4b, b);
    s->x = s->x + s->step;
    out.x = s->x;
    if (s->spherical) {
      out.h = s->a - s->radius;
      out.dh = s->b / s->radius;
    } else {
      out.h = s->a;
      out.dh = s->b;
    }
    return out;
  }
}
