// CPU experiment (no GPU, not part of the product): what would spatial splits (SBVH, Stich / Friedrich / Dietrich 2009) buy on a scene?
// Builds a binary BVH over world triangles twice -- binned SAH with object splits only (what pt_sah.hip / pt_sahdev.h build), and the same with
// spatial splits of OPAQUE triangles (a duplicated opaque reference is harmless under the trace contract: committing the same (t, index) twice changes
// nothing; non-opaque triangles draw random numbers per candidate and are never duplicated) --, collapses both to 4-wide nodes the way the product does
// (children of the child with the largest surface area are pulled up) and counts node visits and triangle tests of closest-hit traversals for camera
// rays and for diffuse bounce rays leaving their hit points.  One triangle per leaf, as in the product.
//   g++ -O2 -fopenmp -std=c++17 tools/sbvh_experiment.cpp -o /tmp/sbvh_experiment ; driven by tools/sbvh_experiment.py
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

struct V3 { double x, y, z; };
static inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
static inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static inline double comp(const V3& v, int a) { return a == 0 ? v.x : a == 1 ? v.y : v.z; }
static inline void setc(V3& v, int a, double s) { (a == 0 ? v.x : a == 1 ? v.y : v.z) = s; }

struct Box {
  V3 lo{1e300, 1e300, 1e300}, hi{-1e300, -1e300, -1e300};
  void grow(V3 p) { lo = {std::min(lo.x, p.x), std::min(lo.y, p.y), std::min(lo.z, p.z)}; hi = {std::max(hi.x, p.x), std::max(hi.y, p.y), std::max(hi.z, p.z)}; }
  void grow(const Box& b) { if(b.valid()) { grow(b.lo); grow(b.hi); } }
  bool valid() const { return lo.x <= hi.x && lo.y <= hi.y && lo.z <= hi.z; }
  double area() const { if(!valid()) return 0; V3 d = hi - lo; return 2 * (d.x * d.y + d.y * d.z + d.z * d.x); }
};
static Box isect(const Box& a, const Box& b)
{
  Box r;
  r.lo = {std::max(a.lo.x, b.lo.x), std::max(a.lo.y, b.lo.y), std::max(a.lo.z, b.lo.z)};
  r.hi = {std::min(a.hi.x, b.hi.x), std::min(a.hi.y, b.hi.y), std::min(a.hi.z, b.hi.z)};
  return r;
}

struct Tri { V3 a, b, c; bool opaque; };
struct Ref { uint32_t tri; Box box; };
struct Node { Box box; int left = -1, right = -1; int tri = -1; int cnt = 0; };  // leaf: tri >= 0 = first entry of g_leafTris, cnt entries
static std::vector<int> g_leafTris;
static size_t           g_maxLeaf = 1;  // argv[4]: references per leaf (the product: 1)

static std::vector<Tri> g_tris;
static const int BINS = 32;

// the part of triangle t inside the slab [lo, hi] of axis a, as a box (Sutherland-Hodgman against the two planes), intersected with `within`
static Box clip_tri_box(const Tri& t, int a, double lo, double hi, const Box& within)
{
  V3 poly[9], tmp[9];
  int n = 3;
  poly[0] = t.a; poly[1] = t.b; poly[2] = t.c;
  for(int pass = 0; pass < 2; ++pass)
  {
    const double plane = pass == 0 ? lo : hi;
    const double sgn   = pass == 0 ? 1.0 : -1.0;
    int m = 0;
    for(int i = 0; i < n; ++i)
    {
      const V3 p = poly[i], q = poly[(i + 1) % n];
      const double dp = sgn * (comp(p, a) - plane), dq = sgn * (comp(q, a) - plane);
      if(dp >= 0) tmp[m++] = p;
      if((dp > 0 && dq < 0) || (dp < 0 && dq > 0))
      {
        V3 r = p + (q - p) * (dp / (dp - dq));
        setc(r, a, plane);
        tmp[m++] = r;
      }
    }
    n = m;
    std::memcpy(poly, tmp, sizeof(V3) * n);
    if(n == 0) break;
  }
  Box b;
  for(int i = 0; i < n; ++i) b.grow(poly[i]);
  return isect(b, within);
}

struct Builder {
  bool   spatial;
  double rootArea = 1, alpha = 1e-5;
  size_t refBudget = 0, refsMade = 0;
  std::vector<Node> nodes;
  size_t splitsSpatial = 0, splitsObject = 0;

  int build(std::vector<Ref>& refs, int depth)
  {
    Node nd;
    for(const Ref& r : refs) nd.box.grow(r.box);
    const int id = int(nodes.size());
    nodes.push_back(nd);
    if(refs.size() <= g_maxLeaf)
    {
      nodes[id].tri = int(g_leafTris.size());
      nodes[id].cnt = int(refs.size());
      for(const Ref& r : refs) g_leafTris.push_back(int(r.tri));
      return id;
    }
    // --- object split: binned SAH over centroids, three axes
    double bestObj = 1e300; int objAxis = -1, objBin = -1;
    Box cb;
    for(const Ref& r : refs) cb.grow((r.box.lo + r.box.hi) * 0.5);
    for(int a = 0; a < 3; ++a)
    {
      const double lo = comp(cb.lo, a), ext = comp(cb.hi, a) - lo;
      if(ext <= 0) continue;
      Box bb[BINS]; size_t cnt[BINS] = {};
      for(const Ref& r : refs)
      {
        int b = std::min(BINS - 1, int((comp((r.box.lo + r.box.hi) * 0.5, a) - lo) / ext * BINS));
        bb[b].grow(r.box); cnt[b]++;
      }
      Box rb[BINS]; size_t rc[BINS];
      Box acc; size_t c = 0;
      for(int b = BINS - 1; b >= 0; --b) { acc.grow(bb[b]); c += cnt[b]; rb[b] = acc; rc[b] = c; }
      acc = Box(); c = 0;
      for(int b = 0; b < BINS - 1; ++b)
      {
        acc.grow(bb[b]); c += cnt[b];
        if(c == 0 || rc[b + 1] == 0) continue;
        const double cost = acc.area() * double(c) + rb[b + 1].area() * double(rc[b + 1]);
        if(cost < bestObj) { bestObj = cost; objAxis = a; objBin = b; }
      }
    }
    // --- spatial split: chopped binning over the node box
    double bestSp = 1e300; int spAxis = -1; double spPlane = 0;
    bool trySpatial = spatial && refsMade < refBudget;
    if(trySpatial && objAxis >= 0)
    {  // only when the object split's children overlap noticeably (Stich et al., eq. 3)
      // children boxes of the best object split
      Box L, R;
      const double lo = comp(cb.lo, objAxis), ext = comp(cb.hi, objAxis) - lo;
      for(const Ref& r : refs)
      {
        int b = std::min(BINS - 1, int((comp((r.box.lo + r.box.hi) * 0.5, objAxis) - lo) / ext * BINS));
        (b <= objBin ? L : R).grow(r.box);
      }
      const Box ov = isect(L, R);
      if(!(ov.valid() && ov.area() / rootArea > alpha))
        trySpatial = false;
    }
    if(trySpatial)
    {
      for(int a = 0; a < 3; ++a)
      {
        const double lo = comp(nd.box.lo, a), ext = comp(nd.box.hi, a) - lo;
        if(ext <= 0) continue;
        Box bb[BINS]; size_t enter[BINS] = {}, leave[BINS] = {};
        const double w = ext / BINS;
        for(const Ref& r : refs)
        {
          int b0 = std::max(0, std::min(BINS - 1, int((comp(r.box.lo, a) - lo) / w)));
          int b1 = std::max(0, std::min(BINS - 1, int((comp(r.box.hi, a) - lo) / w)));
          if(!g_tris[r.tri].opaque)
          {  // never split: the whole reference goes to the bin of its centre
            int bc = std::max(0, std::min(BINS - 1, int((comp((r.box.lo + r.box.hi) * 0.5, a) - lo) / w)));
            bb[bc].grow(r.box); enter[bc]++; leave[bc]++;
            continue;
          }
          for(int b = b0; b <= b1; ++b)
            bb[b].grow(clip_tri_box(g_tris[r.tri], a, lo + w * b, lo + w * (b + 1), r.box));
          enter[b0]++; leave[b1]++;
        }
        Box rb[BINS]; size_t rc[BINS];
        Box acc; size_t c = 0;
        for(int b = BINS - 1; b >= 0; --b) { acc.grow(bb[b]); c += leave[b]; rb[b] = acc; rc[b] = c; }
        acc = Box(); c = 0;
        for(int b = 0; b < BINS - 1; ++b)
        {
          acc.grow(bb[b]); c += enter[b];
          if(c == 0 || rc[b + 1] == 0) continue;
          const double cost = acc.area() * double(c) + rb[b + 1].area() * double(rc[b + 1]);
          if(cost < bestSp) { bestSp = cost; spAxis = a; spPlane = lo + w * (b + 1); }
        }
      }
    }
    std::vector<Ref> Lr, Rr;
    if(spAxis >= 0 && bestSp < bestObj)
    {
      ++splitsSpatial;
      for(const Ref& r : refs)
      {
        const double rl = comp(r.box.lo, spAxis), rh = comp(r.box.hi, spAxis);
        if(rh <= spPlane) Lr.push_back(r);
        else if(rl >= spPlane) Rr.push_back(r);
        else if(!g_tris[r.tri].opaque)
          ((rl + rh) * 0.5 < spPlane ? Lr : Rr).push_back(r);
        else
        {
          Ref a = r, b = r;
          a.box = clip_tri_box(g_tris[r.tri], spAxis, -1e300, spPlane, r.box);
          b.box = clip_tri_box(g_tris[r.tri], spAxis, spPlane, 1e300, r.box);
          if(a.box.valid()) Lr.push_back(a);
          if(b.box.valid()) Rr.push_back(b);
          if(a.box.valid() && b.box.valid()) ++refsMade;
        }
      }
      if(Lr.empty() || Rr.empty() || (Lr.size() == refs.size() && Rr.size() == refs.size()))
      {  // the split separated nothing: fall through to the object split
        Lr.clear(); Rr.clear();
      }
    }
    if(Lr.empty() && objAxis >= 0)
    {
      ++splitsObject;
      const double lo = comp(cb.lo, objAxis), ext = comp(cb.hi, objAxis) - lo;
      for(const Ref& r : refs)
      {
        int b = std::min(BINS - 1, int((comp((r.box.lo + r.box.hi) * 0.5, objAxis) - lo) / ext * BINS));
        (b <= objBin ? Lr : Rr).push_back(r);
      }
    }
    if(Lr.empty() || Rr.empty())
    {  // coincident centroids: halve
      Lr.assign(refs.begin(), refs.begin() + refs.size() / 2);
      Rr.assign(refs.begin() + refs.size() / 2, refs.end());
    }
    { std::vector<Ref>().swap(refs); }
    const int l = build(Lr, depth + 1);
    const int r = build(Rr, depth + 1);
    nodes[id].left = l; nodes[id].right = r;
    return id;
  }
};

// 4-wide collapse: a wide node's children = repeatedly replace the inner child of largest area by its two children until four (the product's k_collapse)
struct Wide { Box box[4]; int child[4]; int tri[4]; int cnt[4]; int n; };
static int collapse(const std::vector<Node>& bn, int root, std::vector<Wide>& out)
{
  const int id = int(out.size());
  out.emplace_back();
  int ch[4] = {bn[root].left, bn[root].right, -1, -1}, n = 2;
  while(n < 4)
  {
    int best = -1; double ba = -1;
    for(int i = 0; i < n; ++i)
      if(bn[ch[i]].tri < 0 && bn[ch[i]].box.area() > ba) { ba = bn[ch[i]].box.area(); best = i; }
    if(best < 0) break;
    const int c = ch[best];
    ch[best] = bn[c].left; ch[n++] = bn[c].right;
  }
  Wide w; w.n = n;
  for(int i = 0; i < n; ++i) { w.box[i] = bn[ch[i]].box; w.tri[i] = bn[ch[i]].tri; w.cnt[i] = bn[ch[i]].cnt; w.child[i] = -1; }
  out[id] = w;
  for(int i = 0; i < n; ++i)
    if(bn[ch[i]].tri < 0)
    {
      const int c = collapse(bn, ch[i], out);
      out[id].child[i] = c;
    }
  return id;
}

struct Ray { V3 o, d; };
static bool hit_box(const Box& b, const Ray& r, const V3& inv, double tmax, double& tn)
{
  double t0 = 0, t1 = tmax;
  for(int a = 0; a < 3; ++a)
  {
    double n = (comp(b.lo, a) - comp(r.o, a)) * comp(inv, a), f = (comp(b.hi, a) - comp(r.o, a)) * comp(inv, a);
    if(n > f) std::swap(n, f);
    t0 = std::max(t0, n); t1 = std::min(t1, f);
  }
  tn = t0;
  return t0 <= t1;
}
static bool hit_tri(const Tri& t, const Ray& r, double& tt)
{
  const V3 e1 = t.b - t.a, e2 = t.c - t.a, p = cross(r.d, e2);
  const double det = dot(e1, p);
  if(std::fabs(det) < 1e-300) return false;
  const double inv = 1 / det;
  const V3 s = r.o - t.a;
  const double u = dot(s, p) * inv;
  if(u < 0 || u > 1) return false;
  const V3 q = cross(s, e1);
  const double v = dot(r.d, q) * inv;
  if(v < 0 || u + v > 1) return false;
  tt = dot(e2, q) * inv;
  return tt > 1e-9;
}
struct Stats { double nodes = 0, tris = 0, rays = 0, leaves = 0; };
static bool trace(const std::vector<Wide>& w, const Ray& r, double& tbest, int& tribest, Stats& st)
{
  const V3 inv{1 / r.d.x, 1 / r.d.y, 1 / r.d.z};
  int stack[256], sp = 0;
  stack[sp++] = 0;
  tbest = 1e300; tribest = -1;
  st.rays += 1;
  while(sp)
  {
    const Wide& n = w[stack[--sp]];
    st.nodes += 1;
    struct C { double t; int i; } c[4]; int nc = 0;
    for(int i = 0; i < n.n; ++i)
    {
      double tn;
      if(hit_box(n.box[i], r, inv, tbest, tn)) c[nc++] = {tn, i};
    }
    std::sort(c, c + nc, [](const C& a, const C& b) { return a.t > b.t; });  // far first on the stack
    for(int k = 0; k < nc; ++k)
    {
      const int i = c[k].i;
      if(n.tri[i] >= 0)
      {
        st.leaves += 1;
        for(int k2 = 0; k2 < n.cnt[i]; ++k2)
        {
          double    tt;
          const int ti = g_leafTris[n.tri[i] + k2];
          st.tris += 1;
          if(hit_tri(g_tris[ti], r, tt) && tt < tbest) { tbest = tt; tribest = ti; }
        }
      }
      else
        stack[sp++] = n.child[i];
    }
  }
  return tribest >= 0;
}

static uint32_t g_rng = 12345u;
static double rnd() { g_rng = g_rng * 747796405u + 2891336453u; uint32_t w = ((g_rng >> ((g_rng >> 28u) + 4u)) ^ g_rng) * 277803737u; w = (w >> 22u) ^ w; return (w >> 8) / 16777216.0; }

int main(int argc, char** argv)
{
  if(argc < 4) { std::fprintf(stderr, "usage: sbvh_experiment tris.bin rays.bin budget\n"); return 2; }
  // tris.bin: uint32 n, then n x (9 float32 + 1 uint32 opaque); rays.bin: uint32 m, then m x 6 float32
  FILE* f = std::fopen(argv[1], "rb"); uint32_t n = 0;
  if(!f || std::fread(&n, 4, 1, f) != 1) return 2;
  g_tris.resize(n);
  for(uint32_t i = 0; i < n; ++i)
  {
    float v[9]; uint32_t op;
    if(std::fread(v, 4, 9, f) != 9 || std::fread(&op, 4, 1, f) != 1) return 2;
    g_tris[i] = {{v[0], v[1], v[2]}, {v[3], v[4], v[5]}, {v[6], v[7], v[8]}, op != 0};
  }
  std::fclose(f);
  f = std::fopen(argv[2], "rb"); uint32_t m = 0;
  if(!f || std::fread(&m, 4, 1, f) != 1) return 2;
  std::vector<Ray> cam(m);
  for(uint32_t i = 0; i < m; ++i)
  {
    float v[6];
    if(std::fread(v, 4, 6, f) != 6) return 2;
    cam[i] = {{v[0], v[1], v[2]}, {v[3], v[4], v[5]}};
  }
  std::fclose(f);
  const double budget = std::atof(argv[3]);
  if(argc > 4) g_maxLeaf = size_t(std::max(1, std::atoi(argv[4])));
  size_t nOpaque = 0;
  for(const Tri& t : g_tris) nOpaque += t.opaque;
  std::printf("%u triangles (%zu opaque), %u camera rays, reference budget +%.0f %%\n", n, nOpaque, m, budget * 100);
  for(int mode = 0; mode < 2; ++mode)
  {
    std::vector<Ref> refs(n);
    Box root;
    for(uint32_t i = 0; i < n; ++i)
    {
      refs[i].tri = i;
      refs[i].box.grow(g_tris[i].a); refs[i].box.grow(g_tris[i].b); refs[i].box.grow(g_tris[i].c);
      root.grow(refs[i].box);
    }
    g_leafTris.clear();
    Builder b;
    b.spatial = mode == 1;
    b.rootArea = root.area();
    b.refBudget = size_t(budget * n);
    b.nodes.reserve(size_t(n) * 3);
    b.build(refs, 0);
    size_t leaves = 0; double sah = 0;
    for(const Node& nd : b.nodes) { if(nd.tri >= 0) ++leaves; sah += nd.box.area() / b.rootArea; }
    std::vector<Wide> wide;
    wide.reserve(b.nodes.size());
    collapse(b.nodes, 0, wide);
    // camera rays, then one diffuse bounce from every hit (cosine-ish: random direction in the hemisphere of the geometric normal facing the ray), then a second
    g_rng = 12345u;
    Stats s0, s1, s2;
    std::vector<Ray> next;
    auto bounce = [&](const std::vector<Ray>& in, Stats& st, std::vector<Ray>& out) {
      out.clear();
      for(const Ray& r : in)
      {
        double t; int tri;
        if(!trace(wide, r, t, tri, st)) continue;
        const Tri& T = g_tris[tri];
        V3 ng = cross(T.b - T.a, T.c - T.a);
        const double l = std::sqrt(dot(ng, ng));
        if(l == 0) continue;
        ng = ng * (1 / l);
        if(dot(ng, r.d) > 0) ng = ng * -1.0;
        V3 d;
        do { d = {rnd() * 2 - 1, rnd() * 2 - 1, rnd() * 2 - 1}; } while(dot(d, d) > 1 || dot(d, d) < 1e-4);
        d = d * (1 / std::sqrt(dot(d, d)));
        d = d + ng;  // cosine-weighted about the normal
        const double dl = std::sqrt(dot(d, d));
        if(dl < 1e-6) continue;
        d = d * (1 / dl);
        out.push_back({r.o + r.d * t + ng * 1e-4, d});
      }
    };
    std::vector<Ray> r1, r2, r3;
    bounce(cam, s0, r1);
    bounce(r1, s1, r2);
    bounce(r2, s2, r3);
    std::printf("%s: %zu leaves (%.3f x triangles), %zu binary nodes, %zu wide nodes, SAH (sum of relative areas) %.1f, spatial / object splits %zu / %zu\n", mode ? "SBVH" : "SAH ", leaves,
                double(leaves) / n, b.nodes.size(), wide.size(), sah, b.splitsSpatial, b.splitsObject);
    std::printf("      camera rays: %.2f wide-node visits, %.2f leaf visits, %.2f triangle tests per ray | bounce 1 (%0.f rays): %.2f / %.2f / %.2f | bounce 2 (%.0f rays): %.2f / %.2f / %.2f\n", s0.nodes / s0.rays,
                s0.leaves / s0.rays, s0.tris / s0.rays, s1.rays, s1.nodes / s1.rays, s1.leaves / s1.rays, s1.tris / s1.rays, s2.rays, s2.nodes / s2.rays, s2.leaves / s2.rays, s2.tris / s2.rays);
  }
  return 0;
}
