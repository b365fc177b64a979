// TEST INFRASTRUCTURE -- part of the CPU oracle (see oracle/README.md). Not linked into the product.
//
// Restatement of shaders/sun_and_sky.glsl (procedural sun & sky, mental-ray physical-sky lineage).
// Inside that file M_PI is the macro 3.1415926535f (sun_and_sky.glsl:23-25), which has the same fp32
// value as the const in globals.glsl.
#pragma once
#include "../include/pt_types.h"
#include "glsl_math.h"

namespace orc {
namespace sky {

static const float SKY_PI = 3.1415926535f;

// :29-32
inline float luminance(vec3 rgb) { return 0.2126f * rgb.x + 0.7152f * rgb.y + 0.0722f * rgb.z; }

// :35-67
inline vec3 xyz2dir(vec3 in_main, float x, float y, float z)
{
  vec3 u, v;
  vec3 omain = in_main;
  if(std::fabs(omain.x) < std::fabs(omain.y))
    u = vec3(0.0f, -omain.z, omain.y);
  else
    u = vec3(omain.z, 0.0f, -omain.x);
  if(length(u) == 0.0f)
  {
    if(std::fabs(in_main.x) < std::fabs(in_main.y))
      u = vec3(0.0f, -in_main.z, in_main.y);
    else
      u = vec3(in_main.z, 0.0f, -in_main.x);
  }
  u = normalize(u);
  v = cross(in_main, u);
  return u * x + v * y + in_main * z;
}

// :70-112
inline vec2 mi_lib_square_to_disk(float inout_r, float inout_phi, float in_x, float in_y)
{
  float local_x = 2 * in_x - 1;
  float local_y = 2 * in_y - 1;
  if(local_x == 0.0f && local_y == 0.0f)
  {
    inout_phi = 0.0f;
    inout_r   = 0.0f;
  }
  else
  {
    if(local_x > -local_y)
    {
      if(local_x > local_y)
      {
        inout_r   = local_x;
        inout_phi = (SKY_PI / 4.0f) * (1.0f + local_y / local_x);
      }
      else
      {
        inout_r   = local_y;
        inout_phi = (SKY_PI / 4.0f) * (3.0f - local_x / local_y);
      }
    }
    else
    {
      if(local_x < local_y)
      {
        inout_r   = -local_x;
        inout_phi = (SKY_PI / 4.0f) * (5.0f + local_y / local_x);
      }
      else
      {
        inout_r   = -local_y;
        inout_phi = (SKY_PI / 4.0f) * (7.0f - local_x / local_y);
      }
    }
  }
  return vec2(inout_r, inout_phi);
}

// :115-135
inline vec3 mi_reflection_dir_diffuse_x(vec3 in_normal, vec2 in_sample)
{
  vec2  r_phi = mi_lib_square_to_disk(0, 0, in_sample.x, in_sample.y);
  float x     = r_phi.x * mcos(r_phi.y);
  float y     = r_phi.x * msin(r_phi.y);
  float z2    = 1.0f - x * x - y * y;
  float z     = z2 > 0.0f ? std::sqrt(z2) : 0.0f;
  return xyz2dir(in_normal, x, y, z);
}

// :138-161
inline vec3 calc_sun_color(vec3 sun_dir, float turbidity)
{
  vec3 sun_color(0.0f);
  vec3 ko(12.0f, 8.5f, 0.9f);
  vec3 wavelength(0.610f, 0.550f, 0.470f);
  vec3 solRad(1.0f * 127500 / 0.9878f, 0.992f * 127500 / 0.9878f, 0.911f * 127500 / 0.9878f);
  if(sun_dir.z > 0.0f)
  {
    float m     = (1.0f / (sun_dir.z + 0.15f * mpow(93.885f - macos(sun_dir.z) * 180 / SKY_PI, -1.253f)));
    float beta  = 0.04608f * turbidity - 0.04586f;
    float alpha = 1.3f;
    vec3  ta    = gexp(gpow(wavelength, vec3(-alpha)) * (-m * beta));
    float l     = 0.0035f;
    vec3  to    = gexp(ko * (-m) * l);
    vec3  tr    = gexp(gpow(wavelength, vec3(-4.08f)) * (-m * 0.008735f));
    sun_color   = tr * ta * to * solRad;
  }
  return sun_color;
}

// :164-219
inline vec3 sky_color_xyz(vec3 in_dir, vec3 in_sun_pos, float in_turbidity, float in_luminance)
{
  vec3  xyz;
  float A, B, C, D, E;
  float cos_gamma = dot(in_sun_pos, in_dir);
  if(cos_gamma > 1.0f)
    cos_gamma = 2.0f - cos_gamma;
  float gamma         = macos(cos_gamma);
  float cos_theta     = in_dir.z;
  float cos_theta_sun = in_sun_pos.z;
  float theta_sun     = macos(cos_theta_sun);
  float t2            = in_turbidity * in_turbidity;
  float ts2           = theta_sun * theta_sun;
  float ts3           = ts2 * theta_sun;
  float zenith_x = ((+0.001650f * ts3 - 0.003742f * ts2 + 0.002088f * theta_sun + 0) * t2
                    + (-0.029028f * ts3 + 0.063773f * ts2 - 0.032020f * theta_sun + 0.003948f) * in_turbidity
                    + (+0.116936f * ts3 - 0.211960f * ts2 + 0.060523f * theta_sun + 0.258852f));
  float zenith_y = ((+0.002759f * ts3 - 0.006105f * ts2 + 0.003162f * theta_sun + 0) * t2
                    + (-0.042149f * ts3 + 0.089701f * ts2 - 0.041536f * theta_sun + 0.005158f) * in_turbidity
                    + (+0.153467f * ts3 - 0.267568f * ts2 + 0.066698f * theta_sun + 0.266881f));
  xyz.y = in_luminance;

  A = -0.019257f * in_turbidity - (0.29f - mpow(cos_theta_sun, 0.5f) * 0.09f);
  B = -0.066513f * in_turbidity + 0.000818f;
  C = -0.000417f * in_turbidity + 0.212479f;
  D = -0.064097f * in_turbidity - 0.898875f;
  E = -0.003251f * in_turbidity + 0.045178f;

  float x = (((1.f + A * mexp(B / cos_theta)) * (1.f + C * mexp(D * gamma) + E * cos_gamma * cos_gamma))
             / ((1 + A * mexp(B / 1.0f)) * (1 + C * mexp(D * theta_sun) + E * cos_theta_sun * cos_theta_sun)));

  A = -0.016698f * in_turbidity - 0.260787f;
  B = -0.094958f * in_turbidity + 0.009213f;
  C = -0.007928f * in_turbidity + 0.210230f;
  D = -0.044050f * in_turbidity - 1.653694f;
  E = -0.010922f * in_turbidity + 0.052919f;

  float y = (((1 + A * mexp(B / cos_theta)) * (1 + C * mexp(D * gamma) + E * cos_gamma * cos_gamma))
             / ((1 + A * mexp(B / 1.0f)) * (1 + C * mexp(D * theta_sun) + E * cos_theta_sun * cos_theta_sun)));

  float local_saturation = 1.0f;
  x = zenith_x * ((x * local_saturation) + (1.0f - local_saturation));
  y = zenith_y * ((y * local_saturation) + (1.0f - local_saturation));

  xyz.x = (x / y) * xyz.y;
  xyz.z = ((1.0f - x - y) / y) * xyz.y;
  return xyz;
}

// :222-250
inline float sky_luminance(vec3 in_dir, vec3 in_sun_pos, float in_turbidity)
{
  float cos_gamma = dot(in_sun_pos, in_dir);
  if(cos_gamma < 0.0f)
    cos_gamma = 0.0f;
  if(cos_gamma > 1.0f)
    cos_gamma = 2.0f - cos_gamma;
  float gamma         = macos(cos_gamma);
  float cos_theta     = in_dir.z;
  float cos_theta_sun = in_sun_pos.z;
  float theta_sun     = macos(cos_theta_sun);

  float A = 0.178721f * in_turbidity - 1.463037f;
  float B = -0.355402f * in_turbidity + 0.427494f;
  float C = -0.022669f * in_turbidity + 5.325056f;
  float D = 0.120647f * in_turbidity - 2.577052f;
  float E = -0.066967f * in_turbidity + 0.370275f;

  return (((1 + A * mexp(B / cos_theta)) * (1 + C * mexp(D * gamma) + E * cos_gamma * cos_gamma))
          / ((1 + A * mexp(B / 1.0f)) * (1 + C * mexp(D * theta_sun) + E * cos_theta_sun * cos_theta_sun)));
}

// :253-267
inline vec3 calc_env_color(vec3 in_sun_dir, vec3 in_dir, float in_turbidity)
{
  float theta_sun = macos(in_sun_dir.z);
  float chi       = (4.0f / 9.0f - in_turbidity / 120.0f) * (SKY_PI - 2 * theta_sun);
  float lum       = 1000.0f * ((4.0453f * in_turbidity - 4.9710f) * mtan(chi) - 0.2155f * in_turbidity + 2.4192f);
  lum *= sky_luminance(in_dir, in_sun_dir, in_turbidity);
  vec3 XYZ = sky_color_xyz(in_dir, in_sun_dir, in_turbidity, lum);
  vec3 env_color(3.241f * XYZ.x - 1.537f * XYZ.y - 0.499f * XYZ.z, -0.969f * XYZ.x + 1.876f * XYZ.y + 0.042f * XYZ.z,
                 0.056f * XYZ.x - 0.204f * XYZ.y + 1.057f * XYZ.z);
  env_color *= SKY_PI;
  return env_color;
}

// :269-289
inline vec3 calc_irrad(vec3 in_data_sun_dir, float in_data_sun_dir_haze)
{
  vec3 colaccu(0.0f);
  vec3 nuState_normal(0.0f, 0.0f, 1.0f);
  for(float u = 1.f / 10.f; u < 1.f; u += 1.f / 5.f)
  {
    for(float v = 1.f / 10.f; v < 1.f; v += 1.f / 5.f)
    {
      vec3 diff = mi_reflection_dir_diffuse_x(nuState_normal, vec2(u, v));
      colaccu += calc_env_color(in_data_sun_dir, diff, in_data_sun_dir_haze);
    }
  }
  colaccu /= 25.0f;
  return colaccu;
}

// :292-309
inline float tweak_saturation(float inout_saturation, float in_haze)
{
  float lowsat = mpow(inout_saturation, 3.0f);
  if(inout_saturation <= 1.0f)
  {
    float local_haze = in_haze;
    local_haze -= 2.0f;
    local_haze /= 15.0f;
    if(local_haze < 0.0f)
      local_haze = 0.0f;
    if(local_haze > 1.0f)
      local_haze = 1.0f;
    local_haze = mpow(local_haze, 3.0f);
    return ((inout_saturation * (1.0f - local_haze)) + lowsat * local_haze);
  }
  return 1.f;
}

// :312-325
inline vec3 arch_vectortweak(vec3 dir, int y_is_up, float horiz_height)
{
  vec3 out_dir = dir;
  if(y_is_up == 1)
    out_dir = vec3(dir.x, dir.z, dir.y);
  if(horiz_height != 0)
  {
    out_dir.z -= horiz_height;
    out_dir = normalize(out_dir);
  }
  return out_dir;
}

// :328-358 (the clamped copy `tint` is dead code in the reference as well)
inline vec3 arch_colortweak(vec3 tint, float saturation, float redness)
{
  float intensity = luminance(tint);
  vec3  out_tint;
  if(saturation <= 0.0f)
    out_tint = vec3(intensity);
  else
    out_tint = tint * saturation + vec3(intensity * (1.0f - saturation));
  out_tint *= vec3(1.0f + redness, 1.f, 1.0f - redness);
  return out_tint;
}

// :361-437
inline vec2 calc_physical_scale(float sun_disk_scale, float sun_glow_intensity, float sun_disk_intensity)
{
  float sun_angular_radius = 0.00465f;
  float sun_disk_radius    = sun_angular_radius * sun_disk_scale;
  float sun_glow_radius    = sun_disk_radius * 10.0f;
  float glow_func_integral = sun_glow_intensity
                             * ((4.f * SKY_PI) - (24.f * SKY_PI) / (sun_glow_radius * sun_glow_radius)
                                + (24.f * SKY_PI) * msin(sun_glow_radius) / (sun_glow_radius * sun_glow_radius * sun_glow_radius));
  float target_sundisk_integral = sun_disk_intensity * SKY_PI;
  float sky_sunglow_scale       = 1.0f;
  float max_glow_integral       = 0.5f * target_sundisk_integral;
  if(glow_func_integral > max_glow_integral)
  {
    sky_sunglow_scale *= max_glow_integral / glow_func_integral;
    target_sundisk_integral -= max_glow_integral;
  }
  else
  {
    target_sundisk_integral -= glow_func_integral;
  }
  float sundisk_area             = 2 * SKY_PI * (1 - mcos(sun_disk_radius));
  float target_sundisk_intensity = target_sundisk_integral / sundisk_area;
  float actual_sundisk_integral  = 1.0f * sundisk_area;
  float actual_sundisk_intensity = sun_disk_intensity * 100.0f * actual_sundisk_integral / sundisk_area;
  return vec2((target_sundisk_intensity == 0.0f) ? 0.0f : target_sundisk_intensity / actual_sundisk_intensity, sky_sunglow_scale);
}

// :440-450
inline float night_brightness_adjustment(vec3 sun_dir)
{
  float lmt = 0.30901699437494742410229341718282f;
  if(sun_dir.z <= -lmt)
    return 0.0f;
  float factor = (sun_dir.z + lmt) / lmt;
  factor *= factor;
  factor *= factor;
  return factor;
}

// :453-599
inline vec3 sun_and_sky(const pt_SunAndSky& ss, vec3 in_direction)
{
  float factor       = 1.0f;
  float night_factor = 1.0f;
  vec3  out_color(0.0f);
  vec3  rgb_scale(ss.rgb_unit_conversion[0], ss.rgb_unit_conversion[1], ss.rgb_unit_conversion[2]);
  vec3  dir          = in_direction;
  float horiz_height = ss.horizon_height / 10.0f;
  dir                = arch_vectortweak(dir, ss.y_is_up, horiz_height);
  float local_haze   = 2.0f + ss.haze;
  if(local_haze < 2.0f)
    local_haze = 2.0f;
  float local_saturation = tweak_saturation(ss.saturation, local_haze);
  if(luminance(rgb_scale) < 0.0f)
    rgb_scale = vec3(1.0f / 80000.0f);
  rgb_scale *= ss.multiplier;
  if(ss.multiplier <= 0.0f)
    return vec3(0);

  float downness = dir.z;
  vec3  real_dir = dir;
  if(dir.z < 0.001f)
  {
    dir.z = 0.001f;
    dir   = normalize(dir);
  }

  vec3 sun_dir(ss.sun_direction[0], ss.sun_direction[1], ss.sun_direction[2]);
  sun_dir           = normalize(sun_dir);
  sun_dir           = arch_vectortweak(sun_dir, ss.y_is_up, horiz_height);
  vec3 real_sun_dir = sun_dir;
  if(sun_dir.z < 0.001f)
  {
    if(sun_dir.z < 0.0f)
      factor = night_brightness_adjustment(sun_dir);
    sun_dir.z = 0.001f;
    sun_dir   = normalize(sun_dir);
  }

  vec3 tint;
  if(factor > 0.0f)
  {
    tint = calc_env_color(sun_dir, dir, local_haze);
    if(factor < 1.0f)
      tint *= factor;
  }
  else
  {
    tint = vec3(0.f);
  }
  vec3 data_sun_color = calc_sun_color(sun_dir, downness > 0 ? local_haze : 2.0f);
  if(ss.sun_disk_intensity > 0.0f && ss.sun_disk_scale > 0.0f)
  {
    float sun_angle  = macos(dot(real_dir, real_sun_dir));
    float sun_radius = 0.00465f * ss.sun_disk_scale * 10.0f;
    if(sun_angle < sun_radius)
    {
      float sky_sundisk_scale = 1.0f;
      float sky_sunglow_scale = 1.0f;
      if(ss.physically_scaled_sun == 1)
      {
        vec2 rv           = calc_physical_scale(ss.sun_disk_scale, ss.sun_glow_intensity, ss.sun_disk_intensity);
        sky_sundisk_scale = rv.x;
        sky_sunglow_scale = rv.y;
      }
      float sun_factor = (1.0f - sun_angle / sun_radius) * 10.0f;
      sun_factor       = (mpow(sun_factor / 10.0f, 3.0f) * 2.0f * ss.sun_glow_intensity * sky_sunglow_scale
                    + gsmoothstep(8.5f, 9.5f + (local_haze / 50.0f), sun_factor) * 100.0f * ss.sun_disk_intensity * sky_sundisk_scale);
      tint += data_sun_color * sun_factor;
    }
  }
  out_color = tint * rgb_scale;
  if(downness <= 0.0f)
  {
    vec3 downcolor(ss.ground_color[0], ss.ground_color[1], ss.ground_color[2]);
    vec3 irrad = calc_irrad(sun_dir, 2.0f);
    downcolor *= (irrad + data_sun_color * sun_dir.z) * rgb_scale;
    if(factor < 1)
      downcolor *= factor;
    float hor_blur = ss.horizon_blur / 10.0f;
    if(hor_blur > 0.0f)
    {
      float dness = -downness;
      dness /= hor_blur;
      if(dness > 1.0f)
        dness = 1.0f;
      dness        = gsmoothstep(0.0f, 1.0f, dness);
      out_color    = out_color * (1.0f - dness) + downcolor * dness;
      night_factor = 1.0f - dness;
    }
    else
    {
      out_color    = downcolor;
      night_factor = 0.0f;
    }
  }

  out_color   = arch_colortweak(out_color, local_saturation, ss.redblueshift);
  vec3 result = out_color;
  if(night_factor > 0.0f)
  {
    vec3 night(ss.night_color[0], ss.night_color[1], ss.night_color[2]);
    night *= night_factor;
    if(result.x < night.x) result.x = night.x;
    if(result.y < night.y) result.y = night.y;
    if(result.z < night.z) result.z = night.z;
  }
  result *= SKY_PI;
  return result;
}

}  // namespace sky
}  // namespace orc
