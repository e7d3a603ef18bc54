"""A float64 evaluation of ONE sample of the jade integrator, written from SURVEY.md section 9 (which distils
PathTrace.cu:686-1474) and NOT from oracle/jade_oracle.c or the HIP code: an independent statement of the
formulas that the closed-form unit tests do not reach - the BSSRDF branch (PathTrace.cu:1029-1178), SSS-diffuse
(:931-1028), direct refraction (:1180-1262, gen_refract_ray :876-894) - plus diffuse, mirror and the pixel
assembly they are embedded in.

Differences from the oracle by construction, so that agreement means something:
  * every quantity is float64 (the oracle and the HIP kernel are fp32 with their own libm), numpy's exp/sqrt/
    sin/cos; only the random numbers are the fp32 values the RNG defines (u = float32(wang) / 2^32);
  * intersection is a brute-force loop over all triangles with the section 9.3 triangle test (no BVH);
  * radiance is unwound Horner-style from explicit (dir, rate) stacks as PathTrace.cu:1410-1413 does
    (the HIP path accumulates forward);
  * the environment must be a constant map (sky(w) = min(c, 10)): the equirect lookup is pinned elsewhere.

Input: the boundary arrays of a HostScene (include/jade_rt.h), i.e. exactly what render_pixel reads."""
import math

import numpy as np

PI = 3.1415926          # #define PI, PathTrace.cu:36
E = 2.71828182846       # Natural_E, PathTrace.cu:37 (a float literal in powf(2.71828182846f, x))
RR = 0.9
SSS = 0.5


def wang_stream(x, y, frame):
    """fshader_render.fsh:82-98; seed of sample s = (x*1973 + y*9277 + (frame+s)*26699) | 1 (jade_rt.h)."""
    s = ((x * 1973 + y * 9277 + frame * 26699) | 1) & 0xffffffff
    while True:
        s = ((s ^ 61) ^ (s >> 16)) & 0xffffffff
        s = (s * 9) & 0xffffffff
        s = s ^ (s >> 4)
        s = (s * 0x27d4eb2d) & 0xffffffff
        s = s ^ (s >> 15)
        yield float(np.float32(s) * np.float32(2.0 ** -32))


class Scene:
    def __init__(self, hs):
        t = hs.a["triangles"]
        f, i = t.view(np.float32).astype(np.float64), t.view(np.int32)
        self.obj = i[:, 0]
        self.p1, self.p2, self.p3 = f[:, 1:4], f[:, 4:7], f[:, 7:10]
        self.norm, self.emis, self.brdf = f[:, 10:13], f[:, 13:16], f[:, 16:19]
        self.reflex, self.refract = i[:, 19], i[:, 20]
        self.rate, self.albedo, self.eta = f[:, 21:24], f[:, 24:27], f[:, 27]
        self.emit = hs.a["emit"]
        self.mapping = hs.a["mapping"]
        self.prefix = hs.a["prefix"].astype(np.float64)
        self.segs = hs.a["segs"]
        env = hs.a["env"].astype(np.float64).reshape(-1, 3)
        assert (env == env[0]).all(), "jade_spec handles constant environments only"
        self.sky = np.minimum(env[0], 10.0)
        self.n = len(t)

    def area(self, k):
        c = np.cross(self.p2[k] - self.p1[k], self.p3[k] - self.p1[k])
        return 0.5 * math.sqrt(c @ c)

    def point(self, k, rx, ry):
        if rx + ry > 1:
            rx, ry = 1 - rx, 1 - ry
        return self.p1[k] + (self.p2[k] - self.p1[k]) * rx + (self.p3[k] - self.p1[k]) * ry

    def hit(self, o, d, skip):
        """Nearest hit over all triangles, section 9.3: (index, point) or (-1, None)."""
        dn = d / math.sqrt(d @ d)
        best, bi, bp = math.inf, -1, None
        for k in range(self.n):
            if k == skip:
                continue
            a, b, c = self.p1[k], self.p2[k], self.p3[k]
            a2 = a - dn * (dn @ (a - o))
            b2 = b - dn * (dn @ (b - o))
            c2 = c - dn * (dn @ (c - o))
            pa, pb, pc = a2 - o, b2 - o, c2 - o
            s1 = dn @ np.cross(pa, pb)
            s2 = dn @ np.cross(pb, pc)
            s3 = dn @ np.cross(pc, pa)
            if not ((s1 > 0 and s2 > 0 and s3 > 0) or (s1 < 0 and s2 < 0 and s3 < 0)):
                continue
            e1, e2, q = b2 - a2, c2 - a2, o - a2
            div = e1[0] * e2[1] - e1[1] * e2[0]
            if div == 0:
                continue
            al = (e2[1] * q[0] - e2[0] * q[1]) / div
            be = (-e1[1] * q[0] + e1[0] * q[1]) / div
            P = a + (b - a) * al + (c - a) * be
            dist = (P - o) @ dn
            if dist > 0 and dist < best:
                best, bi, bp = dist, k, P
        return bi, bp


def sphere_dir(rng):
    c = 2 * (next(rng) - 0.5)
    s = math.sqrt(1 - c * c)
    phi = 2 * PI * next(rng)
    return np.array([s * math.cos(phi), s * math.sin(phi), c])


def refract(I, N, eta):
    """gen_refract_ray, PathTrace.cu:876-894 -> (direction, total internal reflection?)"""
    cosi = I @ N
    if cosi > 0:
        N = -N
    else:
        cosi = -cosi
    c2 = 1.0 - eta * eta * (1.0 - cosi * cosi)
    if c2 > 0:
        return I * eta + N * (eta * cosi - math.sqrt(c2)), False
    return I, True


def emissive(S, k, thr):
    return bool((S.emis[k] > thr).any())


def path_tracing(S, rng, obj, src, out, trace):
    """pathTracing(hit, direction), section 9.5.  `trace` collects the branch taken at every vertex."""
    stack = []
    L = np.zeros(3)
    nE = len(S.emit)
    while len(stack) < 128:
        if emissive(S, obj, 1.4e-5):
            L = S.emis[obj].copy()
            break
        L = np.zeros(3)
        n = S.norm[obj]
        fr = S.brdf[obj] * float(np.float32(1.0 / PI))
        k = 2 if S.refract[obj] != 0 else 1
        u = next(rng)
        if u < 0.5 and S.refract[obj] != 0:
            if S.refract[obj] == 1:
                u2 = next(rng)
                if u2 < SSS:
                    trace.append("sss")
                    fa = S.albedo[obj] * float(np.float32(1.0 / PI))
                    L, nxt = diffuse_like(S, rng, obj, src, out, n, fa, fr, k / SSS)
                else:
                    trace.append("bssrdf")
                    L, nxt = bssrdf(S, rng, obj, src, out, n, k, trace)
            else:
                trace.append("refract")
                res = direct_refraction(S, rng, obj, src, out, n, k)
                if res is None:
                    trace.append("refract-open")
                    return np.zeros(3)          # PathTrace.cu:1231: the whole sample is 0
                L, nxt = res
        elif S.reflex[obj] == 0:
            trace.append("diffuse")
            L, nxt = diffuse_like(S, rng, obj, src, out, n, fr, fr, float(k))
        else:
            trace.append("mirror")
            L, nxt = mirror(S, rng, obj, src, out, n, fr, k)
        if nxt is None:
            break
        push_dir, rate, obj, src, out = nxt
        stack.append((push_dir, rate))
    for d, r in reversed(stack):                # PathTrace.cu:1410-1413
        L = L * r + d
    return L


def diffuse_like(S, rng, obj, src, out, n, f, fr, scale):
    """diffuse (:1266-1364, f = fr, scale = k) and SSS-diffuse (:931-1028, f = albedo/pi, scale = k/0.5)."""
    L = np.zeros(3)
    side = out @ n
    for e in S.emit:
        rx, ry = next(rng), next(rng)
        l = S.point(e, rx, ry) - src
        if (l @ n) * side < 0:
            continue
        h, _ = S.hit(src, l, obj)
        if h == e:
            ll = l @ l
            L = L + S.emis[e] * f * abs((n @ l) * (S.norm[e] @ l)) / ll / ll * S.area(e)
    w = sphere_dir(rng)
    if (w @ n) * side < 0:
        w = -w
    h, _ = S.hit(src, w, obj)
    if h < 0:
        L = L + S.sky * f * abs(n @ w) * 2 * PI
    L = L * scale
    if not next(rng) < RR:
        return L, None
    w = sphere_dir(rng)
    if (w @ n) * side < 0:
        w = -w
    h, hp = S.hit(src, w, obj)
    if h >= 0 and not emissive_ge(S, h):
        w = -w
        rate = fr * abs(w @ n) / RR * scale      # uses fr, also for SSS-diffuse (quirk 9)
        return L, (L, rate, h, hp, w)
    return L, None


def emissive_ge(S, k):
    """the indirect-hit test `all channels < 1.5e-4` negated (PathTrace.cu:1005, 1152, 1341)"""
    return not bool((S.emis[k] < 1.5e-4).all())


def bssrdf(S, rng, obj, src, out, n, k, trace):
    seg = S.segs[S.obj[obj]]
    A = S.prefix[seg[1]]
    xi = next(rng) * A
    left, right, mid = int(seg[0]), int(seg[1]), 0
    while left < right - 1:
        mid = (left + right) // 2
        if xi <= S.prefix[mid]:
            right = mid
        elif xi >= S.prefix[mid]:
            left = mid
    m = int(S.mapping[mid])                      # the LAST mid tried (0 if the loop never ran), quirk 10
    rx, ry = next(rng), next(rng)
    Q = S.point(m, rx, ry)
    inner = Q - src
    r = math.sqrt(inner @ inner)
    d = S.rate[m]
    Rd = (np.exp(-r / d) + np.exp(-r / 3.0 / d)) / (d * (8 * PI * r))
    eta = S.eta[m]
    R0 = ((eta - 1) / (eta + 1)) ** 2
    Fi = R0 + (1 - R0) * (1 - abs(n @ out)) ** 5
    Rd = Rd * Fi
    nm = S.norm[m]
    Aobj = S.prefix[S.segs[S.obj[m]][1]]
    if abs(inner @ nm) < 1e-5 * r:
        # src and Q in one plane (same flat face): the two side tests below, dot(w, n_m) * dot(Q - src, n_m) <> 0
        # (PathTrace.cu:1115, 1140), take the sign of rounding noise - the reference's result is then arbitrary
        # and no two precisions agree on it.  Flagged so that a comparison can leave the sample out.
        trace.append("bssrdf-coplanar")

    def Fo(v):
        return R0 - (1 - R0) * (1 - abs(v @ nm)) ** 5      # the minus sign as written (quirk 8)

    L = np.zeros(3)
    for e in S.emit:
        ex, ey = next(rng), next(rng)
        l = S.point(e, ex, ey) - Q
        h, _ = S.hit(Q, l, m)
        if h == e:
            ll = l @ l
            L = L + S.emis[e] * Fo(l / math.sqrt(ll)) * Rd * abs((nm @ l) * (S.norm[e] @ l)) / ll / ll * S.area(e) / PI * Aobj
    w = sphere_dir(rng)
    if (w @ nm) * (inner @ nm) < 0:
        w = -w
    h, _ = S.hit(Q, w, m)
    if h < 0:
        L = L + S.sky * Fo(w) * Rd * abs(nm @ w) * 2
    L = L * (k / (1 - SSS))
    w = sphere_dir(rng)                          # drawn BEFORE the roulette draw (:1136-1145)
    if (w @ nm) * (inner @ nm) > 0:
        w = -w
    if not next(rng) < RR:
        return L, None
    h, hp = S.hit(Q, w, m)
    if h >= 0 and not emissive_ge(S, h):
        w = -w
        rate = Rd * Fo(w) * abs(w @ nm) * Aobj * 2 / RR * (k / (1 - SSS))
        return L, (L, rate, h, hp, w)
    return L, None


def direct_refraction(S, rng, obj, src, out, n, k):
    eta = S.eta[obj]
    R0 = ((1 - eta) / (1 + eta)) ** 2
    Fi = R0 + (1 - R0) * (1 - abs(n @ out)) ** 5
    t, _ = refract(-out, n, float(np.float32(1.0 / eta)))
    rate = np.full(3, 1 - Fi)
    origin, prev = src, obj
    for _ in range(32):
        h, hp = S.hit(origin, t, prev)
        if h < 0:
            return None
        nh = S.norm[h]
        t, tir = refract(t, nh, eta)
        seg = origin - hp
        rate = rate * S.rate[h] ** math.sqrt(seg @ seg)
        origin, prev = hp, h
        Fo = R0 - (1 - R0) * (1 - abs(t @ nh)) ** 5
        u = next(rng)
        if tir or u < 0.2:
            t = t - nh * (2 * (t @ nh))
            if not tir:
                rate = rate * (Fo * 5)
        else:
            rate = rate * ((1.0 - Fo) * 1.25)
            break
    if not next(rng) < RR:
        return np.zeros(3), None
    h, hp = S.hit(origin, t, prev)
    if h >= 0:                                    # emitters included: the next iteration's first test ends the path
        return np.zeros(3), (np.zeros(3), rate * (k / RR), h, hp, -t)
    return S.sky * rate * (k / RR), None


def mirror(S, rng, obj, src, out, n, fr, k):
    if not next(rng) < RR:
        return np.zeros(3), None
    r = n * (2 * (out @ n)) - out
    kk = k / (RR / PI)
    h, hp = S.hit(src, r, obj)
    if h >= 0:
        return np.zeros(3), (np.zeros(3), fr * kk, h, hp, -r)
    return S.sky * fr * kk, None


def sample(S, x, y, width, height, eye, cam, frame, trace=None):
    """One sample of pixel (x, y): section 9.2 camera ray, 9.6 pixel assembly."""
    rng = wang_stream(x, y, frame)
    trace = trace if trace is not None else []
    lx = (-1 + 2.0 / width * (x + next(rng) - 0.5)) * (width / height)
    ly = -1 + 2.0 / height * (y + next(rng) - 0.5)
    M = np.asarray(cam, np.float64).reshape(4, 4)            # M[col][row]
    v = np.array([lx, ly, -1.5, 0.0])
    d = np.array([sum(M[c][r] * v[c] for c in range(4)) for r in range(3)])
    d = d / math.sqrt(d @ d)
    o = np.asarray(eye, np.float64)
    h, hp = S.hit(o, d, -1)
    if h < 0:
        trace.append("sky")
        return S.sky.copy()
    return S.emis[h] + path_tracing(S, rng, h, hp, -d, trace)
