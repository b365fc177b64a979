// Procedural sun & sky for gfx950 (used when SunAndSky::in_use == 1).
// Behavioural contract: shaders/sun_and_sky.glsl:31-601 (Preetham-style sky, sun disk + glow, ground
// blend below the horizon incl. the 25-sample irradiance estimate, night colour).  Arithmetic order
// follows the reference expression by expression; only libm-class functions differ in ulps.
#pragma once
#include "pt_math.h"
#include "../../include/pt_types.h"

#define SKY_PI 3.1415926535f

PT_DEV float sky_lum(f3 c) { return 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z; }

PT_DEV f3 sky_frame_dir(f3 n, float x, float y, float z)  // xyz2dir :35-67
{
  f3 u = (fabsf(n.x) < fabsf(n.y)) ? f3{0.0f, -n.z, n.y} : f3{n.z, 0.0f, -n.x};
  // (the reference's "degenerate transform" retry recomputes the same vector)
  u    = unit(u);
  f3 v = cross3(n, u);
  return u * x + v * y + n * z;
}

PT_DEV f2 sky_square_to_disk(float in_x, float in_y)  // :70-112
{
  float lx = 2 * in_x - 1, ly = 2 * in_y - 1;
  float r = 0.0f, phi = 0.0f;
  if(!(lx == 0.0f && ly == 0.0f))
  {
    if(lx > -ly)
    {
      if(lx > ly)
      {
        r   = lx;
        phi = (SKY_PI / 4.0f) * (1.0f + ly / lx);
      }
      else
      {
        r   = ly;
        phi = (SKY_PI / 4.0f) * (3.0f - lx / ly);
      }
    }
    else
    {
      if(lx < ly)
      {
        r   = -lx;
        phi = (SKY_PI / 4.0f) * (5.0f + ly / lx);
      }
      else
      {
        r   = -ly;
        phi = (SKY_PI / 4.0f) * (7.0f - lx / ly);
      }
    }
  }
  return f2{r, phi};
}

PT_DEV f3 sky_sun_color(f3 sun_dir, float turbidity)  // calc_sun_color :138-161
{
  f3 c = splat3(0.0f);
  if(sun_dir.z > 0.0f)
  {
    const f3 ko     = f3{12.0f, 8.5f, 0.9f};
    const f3 wl     = f3{0.610f, 0.550f, 0.470f};
    const f3 solRad = f3{1.0f * 127500 / 0.9878f, 0.992f * 127500 / 0.9878f, 0.911f * 127500 / 0.9878f};
    float    m      = (1.0f / (sun_dir.z + 0.15f * pt_pow(93.885f - pt_acos(sun_dir.z) * 180 / SKY_PI, -1.253f)));
    float    beta   = 0.04608f * turbidity - 0.04586f;
    f3       ta     = exp3(pow3(wl, -1.3f) * (-m * beta));
    f3       to     = exp3(ko * (-m) * 0.0035f);
    f3       tr     = exp3(pow3(wl, -4.08f) * (-m * 0.008735f));
    c               = tr * ta * to * solRad;
  }
  return c;
}

PT_DEV float sky_perez(float A, float B, float C, float D, float E, float cos_theta, float gamma, float cos_gamma, float theta_sun, float cos_theta_sun)
{
  return (((1 + A * pt_exp(B / cos_theta)) * (1 + C * pt_exp(D * gamma) + E * cos_gamma * cos_gamma))
          / ((1 + A * pt_exp(B / 1.0f)) * (1 + C * pt_exp(D * theta_sun) + E * cos_theta_sun * cos_theta_sun)));
}

PT_DEV f3 sky_color_xyz(f3 dir, f3 sun, float T, float lum)  // :164-219
{
  float cos_gamma = dot3(sun, dir);
  if(cos_gamma > 1.0f)
    cos_gamma = 2.0f - cos_gamma;
  float gamma = pt_acos(cos_gamma);
  float ct = dir.z, cts = sun.z;
  float ts  = pt_acos(cts);
  float t2  = T * T;
  float ts2 = ts * ts;
  float ts3 = ts2 * ts;
  float zx  = ((+0.001650f * ts3 - 0.003742f * ts2 + 0.002088f * ts + 0) * t2 + (-0.029028f * ts3 + 0.063773f * ts2 - 0.032020f * ts + 0.003948f) * T
              + (+0.116936f * ts3 - 0.211960f * ts2 + 0.060523f * ts + 0.258852f));
  float zy  = ((+0.002759f * ts3 - 0.006105f * ts2 + 0.003162f * ts + 0) * t2 + (-0.042149f * ts3 + 0.089701f * ts2 - 0.041536f * ts + 0.005158f) * T
              + (+0.153467f * ts3 - 0.267568f * ts2 + 0.066698f * ts + 0.266881f));
  float A = -0.019257f * T - (0.29f - pt_pow(cts, 0.5f) * 0.09f);
  float B = -0.066513f * T + 0.000818f;
  float C = -0.000417f * T + 0.212479f;
  float D = -0.064097f * T - 0.898875f;
  float E = -0.003251f * T + 0.045178f;
  float x = sky_perez(A, B, C, D, E, ct, gamma, cos_gamma, ts, cts);
  A       = -0.016698f * T - 0.260787f;
  B       = -0.094958f * T + 0.009213f;
  C       = -0.007928f * T + 0.210230f;
  D       = -0.044050f * T - 1.653694f;
  E       = -0.010922f * T + 0.052919f;
  float y = sky_perez(A, B, C, D, E, ct, gamma, cos_gamma, ts, cts);
  const float sat = 1.0f;
  x       = zx * ((x * sat) + (1.0f - sat));
  y       = zy * ((y * sat) + (1.0f - sat));
  f3 xyz_;
  xyz_.y = lum;
  xyz_.x = (x / y) * xyz_.y;
  xyz_.z = ((1.0f - x - y) / y) * xyz_.y;
  return xyz_;
}

PT_DEV float sky_luminance(f3 dir, f3 sun, float T)  // :222-250
{
  float cos_gamma = dot3(sun, dir);
  if(cos_gamma < 0.0f)
    cos_gamma = 0.0f;
  if(cos_gamma > 1.0f)
    cos_gamma = 2.0f - cos_gamma;
  float gamma = pt_acos(cos_gamma);
  float ts    = pt_acos(sun.z);
  float A     = 0.178721f * T - 1.463037f;
  float B     = -0.355402f * T + 0.427494f;
  float C     = -0.022669f * T + 5.325056f;
  float D     = 0.120647f * T - 2.577052f;
  float E     = -0.066967f * T + 0.370275f;
  return sky_perez(A, B, C, D, E, dir.z, gamma, cos_gamma, ts, sun.z);
}

PT_DEV f3 sky_env_color(f3 sun, f3 dir, float T)  // calc_env_color :253-267
{
  float ts  = pt_acos(sun.z);
  float chi = (4.0f / 9.0f - T / 120.0f) * (SKY_PI - 2 * ts);
  float lum = 1000.0f * ((4.0453f * T - 4.9710f) * pt_tan(chi) - 0.2155f * T + 2.4192f);
  lum *= sky_luminance(dir, sun, T);
  f3 X = sky_color_xyz(dir, sun, T, lum);
  f3 c = f3{3.241f * X.x - 1.537f * X.y - 0.499f * X.z, -0.969f * X.x + 1.876f * X.y + 0.042f * X.z, 0.056f * X.x - 0.204f * X.y + 1.057f * X.z};
  c *= SKY_PI;
  return c;
}

PT_DEV f3 sky_irradiance(f3 sun, float haze)  // calc_irrad :269-289
{
  f3       acc = splat3(0.0f);
  const f3 up  = f3{0.0f, 0.0f, 1.0f};
  for(float u = 1.f / 10.f; u < 1.f; u += 1.f / 5.f)
  {
    for(float v = 1.f / 10.f; v < 1.f; v += 1.f / 5.f)
    {
      f2    rp = sky_square_to_disk(u, v);
      float x  = rp.x * pt_cos(rp.y);
      float y  = rp.x * pt_sin(rp.y);
      float z2 = 1.0f - x * x - y * y;
      float z  = z2 > 0.0f ? sqrtf(z2) : 0.0f;
      acc += sky_env_color(sun, sky_frame_dir(up, x, y, z), haze);
    }
  }
  acc /= 25.0f;
  return acc;
}

PT_DEV f3 sky_tweak_dir(f3 dir, int y_is_up, float horiz_height)  // arch_vectortweak :312-325
{
  f3 o = dir;
  if(y_is_up == 1)
    o = f3{dir.x, dir.z, dir.y};
  if(horiz_height != 0)
  {
    o.z -= horiz_height;
    o = unit(o);
  }
  return o;
}

PT_DEV f2 sky_physical_scale(float disk_scale, float glow_intensity, float disk_intensity)  // :361-437
{
  float disk_radius   = 0.00465f * disk_scale;
  float glow_radius   = disk_radius * 10.0f;
  float glow_integral = glow_intensity
                        * ((4.f * SKY_PI) - (24.f * SKY_PI) / (glow_radius * glow_radius) + (24.f * SKY_PI) * pt_sin(glow_radius) / (glow_radius * glow_radius * glow_radius));
  float target       = disk_intensity * SKY_PI;
  float glow_scale   = 1.0f;
  float max_glow     = 0.5f * target;
  if(glow_integral > max_glow)
  {
    glow_scale *= max_glow / glow_integral;
    target -= max_glow;
  }
  else
  {
    target -= glow_integral;
  }
  float area             = 2 * SKY_PI * (1 - pt_cos(disk_radius));
  float target_intensity = target / area;
  float actual_integral  = 1.0f * area;
  float actual_intensity = disk_intensity * 100.0f * actual_integral / area;
  return f2{(target_intensity == 0.0f) ? 0.0f : target_intensity / actual_intensity, glow_scale};
}

PT_DEV f3 sun_and_sky(const pt_SunAndSky& ss, f3 in_direction)  // :453-599
{
  float factor = 1.0f, night_factor = 1.0f;
  f3    rgb_scale    = f3{ss.rgb_unit_conversion[0], ss.rgb_unit_conversion[1], ss.rgb_unit_conversion[2]};
  float horiz_height = ss.horizon_height / 10.0f;
  f3    dir          = sky_tweak_dir(in_direction, ss.y_is_up, horiz_height);
  float haze         = 2.0f + ss.haze;
  if(haze < 2.0f)
    haze = 2.0f;
  // tweak_saturation :292-309
  float saturation = 1.f;
  {
    float lowsat = pt_pow(ss.saturation, 3.0f);
    if(ss.saturation <= 1.0f)
    {
      float h = haze;
      h -= 2.0f;
      h /= 15.0f;
      if(h < 0.0f)
        h = 0.0f;
      if(h > 1.0f)
        h = 1.0f;
      h          = pt_pow(h, 3.0f);
      saturation = ((ss.saturation * (1.0f - h)) + lowsat * h);
    }
  }
  if(sky_lum(rgb_scale) < 0.0f)
    rgb_scale = splat3(1.0f / 80000.0f);
  rgb_scale *= ss.multiplier;
  if(ss.multiplier <= 0.0f)
    return splat3(0.0f);

  float downness = dir.z;
  f3    real_dir = dir;
  if(dir.z < 0.001f)
  {
    dir.z = 0.001f;
    dir   = unit(dir);
  }
  f3 sun      = unit(f3{ss.sun_direction[0], ss.sun_direction[1], ss.sun_direction[2]});
  sun         = sky_tweak_dir(sun, ss.y_is_up, horiz_height);
  f3 real_sun = sun;
  if(sun.z < 0.001f)
  {
    if(sun.z < 0.0f)
    {
      // night_brightness_adjustment :440-450
      const float lmt = 0.30901699437494742410229341718282f;
      if(sun.z <= -lmt)
        factor = 0.0f;
      else
      {
        float f = (sun.z + lmt) / lmt;
        f *= f;
        f *= f;
        factor = f;
      }
    }
    sun.z = 0.001f;
    sun   = unit(sun);
  }

  f3 tint;
  if(factor > 0.0f)
  {
    tint = sky_env_color(sun, dir, haze);
    if(factor < 1.0f)
      tint *= factor;
  }
  else
    tint = splat3(0.f);

  f3 sun_color = sky_sun_color(sun, downness > 0 ? haze : 2.0f);
  if(ss.sun_disk_intensity > 0.0f && ss.sun_disk_scale > 0.0f)
  {
    float sun_angle  = pt_acos(dot3(real_dir, real_sun));
    float sun_radius = 0.00465f * ss.sun_disk_scale * 10.0f;
    if(sun_angle < sun_radius)
    {
      float disk_scale = 1.0f, glow_scale = 1.0f;
      if(ss.physically_scaled_sun == 1)
      {
        f2 rv      = sky_physical_scale(ss.sun_disk_scale, ss.sun_glow_intensity, ss.sun_disk_intensity);
        disk_scale = rv.x;
        glow_scale = rv.y;
      }
      float sf = (1.0f - sun_angle / sun_radius) * 10.0f;
      sf       = (pt_pow(sf / 10.0f, 3.0f) * 2.0f * ss.sun_glow_intensity * glow_scale + smooth(8.5f, 9.5f + (haze / 50.0f), sf) * 100.0f * ss.sun_disk_intensity * disk_scale);
      tint += sun_color * sf;
    }
  }
  f3 out_color = tint * rgb_scale;
  if(downness <= 0.0f)
  {
    f3 down  = f3{ss.ground_color[0], ss.ground_color[1], ss.ground_color[2]};
    f3 irrad = sky_irradiance(sun, 2.0f);
    down *= (irrad + sun_color * sun.z) * rgb_scale;
    if(factor < 1)
      down *= factor;
    float blur = ss.horizon_blur / 10.0f;
    if(blur > 0.0f)
    {
      float dn = -downness;
      dn /= blur;
      if(dn > 1.0f)
        dn = 1.0f;
      dn           = smooth(0.0f, 1.0f, dn);
      out_color    = out_color * (1.0f - dn) + down * dn;
      night_factor = 1.0f - dn;
    }
    else
    {
      out_color    = down;
      night_factor = 0.0f;
    }
  }
  // arch_colortweak :328-358
  {
    float intensity = sky_lum(out_color);
    f3    t;
    if(saturation <= 0.0f)
      t = splat3(intensity);
    else
      t = out_color * saturation + splat3(intensity * (1.0f - saturation));
    t *= f3{1.0f + ss.redblueshift, 1.f, 1.0f - ss.redblueshift};
    out_color = t;
  }
  f3 result = out_color;
  if(night_factor > 0.0f)
  {
    f3 night = f3{ss.night_color[0], ss.night_color[1], ss.night_color[2]};
    night *= night_factor;
    if(result.x < night.x) result.x = night.x;
    if(result.y < night.y) result.y = night.y;
    if(result.z < night.z) result.z = night.z;
  }
  result *= SKY_PI;
  return result;
}
