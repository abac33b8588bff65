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
  return (1.0 + v * v) * dn / n;
}

static double accel_sph(const oracle_env_atm* atm, double radius, double r, double v) {
  double h = r - radius;
  double n = oracle_n(atm, h);
  double dn = oracle_dn(atm, h);
  return r + 2.0 * v * v / r + (r * r + v * v) * dn / n;
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
    k2a = b + half * k1b;
    k2b = ACC(a + half * k1a, k2a);
    k3a = b + half * k2b;
    k3b = ACC(a + half * k2a, k3a);
    k4a = b + d * k3b;
    k4b = ACC(a + d * k3a, k4a);
#undef ACC
    s->a = a + sixth * (k1a + 2.0 * k2a + 2.0 * k3a + k4a);
    s->b = b + sixth * (k1b + 2.0 * k2b + 2.0 * k3b + k4b);
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
