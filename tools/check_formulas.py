#!/usr/bin/env python3
"""Symbolic identity of every FORMULA of the hot path with the reference's source text.

tools/check_literals.py proves the coefficient tables of the Bessel expansions and of the Heyvaerts elements identical to
the reference's; everything else that is text in the reference -- the kinematics of gamma_integrand and both branches of
its stabilised gamma sin(xi) (symphony.rs:398-479), the gamma limits / rel_width / lobe split (312-363), both
prefactors (173-183), fill_coord_vars and dfdsigma (heyvaerts.rs:194-201, 472-493), the limits of the inner integrals
(213-296) and the final scaling (189-190), calc_f / calc_f_derivatives of the four distributions (power_law.rs:36-62,
thermal_juettner.rs:29-39, pitchy_pl.rs:32-64, pitchy_kappa.rs:38-62) with their normalisation integrands, the cgs
constants and the scaling of compute_cgs (lib.rs:55-67, 163-173) -- was pinned only through 1 % fixtures and
finite-difference tests.  With no Rust toolchain, this is the only pin tighter than that:

  * the reference's statements are taken from the .rs files as TEXT, Rust method syntax (x.sqrt(), x.powi(2), ..) is
    rewritten to function calls, every literal becomes an exact rational, and the statements are executed in order on
    sympy symbols;
  * the statements of rimphony_amd/csrc/*.h (the kernels' device functions) and of oracle/*.c are executed the same way
    (rim_sqrt -> sqrt, rim_div_moderate(a, b) -> a / b, powexp_normal(x, y, e) -> x^y exp(e), ...: the restricted /
    reformulated leaf functions are mapped to the mathematical function they evaluate -- THEIR accuracy is
    tests/test_detmath.py's subject);
  * each named quantity must be algebraically IDENTICAL on the three sides (sympy: the difference simplifies to 0).

Runs where /root/reference is mounted (build container only; reads it as data).  Exit status 0 = all identical.
RIMPHONY_CHECK_DEV_SYMPHONY / _DEV_HEYVAERTS / _ORACLE_DIST: a file to read in place of dev_symphony.h / dev_heyvaerts.h /
rimo_dist.c (the test suite feeds mutated copies to show that one flipped sign in each is caught).
usage: python tools/check_formulas.py [/root/reference]"""
import os
import re
import sys

import sympy as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
NUM = re.compile(r"(?<![\w.])(\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?)(?![\w(])")
METHODS = "sqrt|powi|powf|abs|exp|ln|sin|cos|min|max"
failures = []
rows = []


def exact(text):
    return NUM.sub(lambda m: 'R("%s")' % m.group(1), text)


def rust_methods_to_calls(t):
    """x.sqrt() -> M_sqrt(x), (a + b).powi(2) -> M_powi((a + b), 2), f(x).abs() -> M_abs(f(x)): innermost first."""
    pat = re.compile(r"\.(%s)\(" % METHODS)
    while True:
        m = pat.search(t)
        if not m:
            return t
        end_recv = m.start()
        i = end_recv - 1
        if t[i] == ")":
            depth = 0
            while i >= 0:
                if t[i] == ")":
                    depth += 1
                elif t[i] == "(":
                    depth -= 1
                    if depth == 0:
                        break
                i -= 1
            i -= 1
            while i >= 0 and (t[i].isalnum() or t[i] in "_."):
                i -= 1
            start = i + 1
        else:
            while i >= 0 and (t[i].isalnum() or t[i] in "_."):
                i -= 1
            start = i + 1
        j = m.end()
        depth = 1
        while depth:
            if t[j] == "(":
                depth += 1
            elif t[j] == ")":
                depth -= 1
            j += 1
        recv, args = t[start:end_recv], t[m.end():j - 1].strip()
        t = t[:start] + "M_%s(%s%s)" % (m.group(1), recv, (", " + args) if args else "") + t[j:]


def strip_comments(t):
    t = re.sub(r"/\*.*?\*/", " ", t, flags=re.S)
    return re.sub(r"//[^\n]*", "", t)


def between(text, start, stop):
    i = text.index(start)
    j = text.index(stop, i + len(start))
    return text[i:j]


def fn_body(text, head):
    """Text of the brace block that follows `head`."""
    i = text.index(head)
    i = text.index("{", i)
    depth, j = 0, i
    while True:
        if text[j] == "{":
            depth += 1
        elif text[j] == "}":
            depth -= 1
            if depth == 0:
                return text[i + 1:j]
        j += 1


def normalise(body, lang):
    body = strip_comments(body)
    body = re.sub(r"RIM_(PROF|HIT|LANES)\w*\([^;]*\);", "", body)
    if lang == "rust":
        body = body.replace("f64::consts::PI", "PI")
        body = rust_methods_to_calls(body)
        body = body.replace("self.", "")
    else:
        body = re.sub(r"\b(M|RimMath<\w+>)::", "", body)
        body = re.sub(r"<\s*(KIND|PREC|0|1)(\s*,\s*(KIND|PREC|0|1))*\s*>\s*\(", "(", body)
        body = re.sub(r"\b(st|pt|c|d|so|sh|T|hc)\s*(->|\.)\s*", "", body)
        body = re.sub(r"\bpar\[(\d)\]", r"par\1", body)
        body = re.sub(r"\b(case\s+\w+|default)\s*:", "", body)
    return body


def run(body, env, lang, only=None):
    """Execute `let x = e;` / `const double x = e;` / `x = e;` statements in order on sympy values.  Statements that
    cannot be evaluated (calls into other code, control flow) are skipped; a later use of their name then fails
    loudly in same()."""
    e = dict(env)
    e["R"] = sp.Rational
    stmts, depth, cur = [], 0, ""
    for ch in normalise(body, lang):
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
        if ch == ";" and depth <= 0:
            stmts.append(cur)
            cur = ""
            depth = 0
        else:
            cur += ch
    stmts.append(cur)
    for st in stmts:
        st = st.strip().replace("\n", " ")
        st = re.split(r"[{}]", st)[-1].strip()
        mm = re.match(r"^(?:let\s+(?:mut\s+)?|const\s+double\s+|double\s+)?(\w+)\s*=\s*(?!=)(.*)$", st, re.S)
        if not mm:
            continue
        name, rhs = mm.group(1), mm.group(2).strip()
        # `const double a = X, b = Y`: one declaration, several names
        parts = re.split(r",\s*(?=\w+\s*=(?!=))", rhs) if lang != "rust" and not re.search(r"\([^()]*,\s*\w+\s*=", rhs) else [rhs]
        pairs = [(name, parts[0])]
        for extra in parts[1:]:
            m2 = re.match(r"(\w+)\s*=\s*(.*)$", extra, re.S)
            pairs.append((m2.group(1), m2.group(2)))
        for nm, rh in pairs:
            if only and nm not in only:
                continue
            try:
                e[nm] = eval(exact(rh.strip()), {"__builtins__": {}}, e)
            except Exception:
                e.pop(nm, None)
    return e


def same(name, ours, ref, where):
    try:
        d = sp.simplify(sp.together(sp.expand_power_base(ours - ref, force=True)))
        if d != 0:
            d = sp.simplify(sp.powsimp(sp.expand(d), force=True))
        ok = d == 0
    except Exception as ex:                      # a missing quantity on either side
        ok, d = False, ex
    rows.append("%-74s %s" % ("%s  [%s]" % (name, where), "identical" if ok else "DIFFERENT"))
    print(rows[-1])
    if not ok:
        failures.append(name + " " + where)


def pos(names):
    return sp.symbols(names, positive=True)


def base_env(extra=None):
    f = {"M_sqrt": sp.sqrt, "rim_sqrt": sp.sqrt, "m_sqrt": sp.sqrt, "sqrt": sp.sqrt, "__builtin_sqrt": sp.sqrt,
         "M_powi": lambda a, n: a ** n, "M_powf": lambda a, b: a ** b, "M_abs": sp.Abs, "rim_fabs": sp.Abs, "m_fabs": sp.Abs,
         "M_exp": sp.exp, "rim_exp": sp.exp, "m_exp": sp.exp, "exp": sp.exp, "M_ln": sp.log, "rim_log": sp.log, "m_log": sp.log,
         "M_min": sp.Min, "rust_min": sp.Min, "rim_div_moderate": lambda a, b: a / b,
         "rim_pow": lambda a, b: a ** b, "m_pow": lambda a, b: a ** b, "pow": lambda a, b: a ** b,
         "rim_pow15": lambda a: a ** sp.Rational(3, 2), "m_pow15": lambda a: a ** sp.Rational(3, 2),
         "rim_pow43": lambda a: a ** sp.Rational(4, 3), "m_pow43": lambda a: a ** sp.Rational(4, 3),
         "powexp_normal": lambda x, y, ex: x ** y * sp.exp(ex), "rim_powexp_normal": lambda x, y, ex: x ** y * sp.exp(ex),
         "POWEXP": lambda x, y, ex: x ** y * sp.exp(ex), "PI": sp.pi, "RIM_PI": sp.pi}
    if extra:
        f.update(extra)
    return f


def read(*parts, env_override=None):
    path = os.environ.get(env_override) if env_override else None
    return open(path or os.path.join(*parts)).read()


def literal_only(text):
    """The oracle sources choose between the literal form of a quantity and the deterministic (lock-step) one with
    RIMO_LIT(bit) (oracle/rimo_math.h: compile-time constants in the two shipped flavours, run-time switches in the
    attribution build).  What is compared with the reference's text here is the LITERAL branch of each such choice."""
    text = re.sub(r"if \(!RIMO_LIT\(\w+\)\) \{[^{}]*\}", "", text)                                   # a deterministic-only block
    text = re.sub(r"if \(RIMO_LIT\(\w+\)\)\s*(return [^;]+;)\s*return [^;]+;", r"\1", text)          # if (lit) return A; return B;
    text = re.sub(r"if \(RIMO_LIT\(\w+\)\)\s*([^;{}]+;)\s*else\s*[^;{}]+;", r"\1", text)           # if (lit) A; else B;
    text = re.sub(r"RIMO_LIT\(\w+\) \? ([^;:\n]+) : [^;\n]+;", r"\1;", text)                              # lit ? A : B
    return text


def main():
    if not os.path.isdir(REF):
        print("reference not mounted at %s: nothing to check" % REF)
        return 0
    sym_rs = read(REF, "src", "symphony.rs")
    hey_rs = read(REF, "src", "heyvaerts.rs")
    lib_rs = read(REF, "src", "lib.rs")
    dev_sym = read(ROOT, "rimphony_amd", "csrc", "dev_symphony.h", env_override="RIMPHONY_CHECK_DEV_SYMPHONY")
    wave_sym = read(ROOT, "rimphony_amd", "csrc", "symphony_wave.h")
    dev_hey = read(ROOT, "rimphony_amd", "csrc", "dev_heyvaerts.h", env_override="RIMPHONY_CHECK_DEV_HEYVAERTS")
    wave_hey = read(ROOT, "rimphony_amd", "csrc", "heyvaerts_wave.h")
    ora_sym = read(ROOT, "oracle", "rimo_symphony.c")
    ora_hey = literal_only(read(ROOT, "oracle", "rimo_heyvaerts.c"))
    ora_dist = literal_only(read(ROOT, "oracle", "rimo_dist.c", env_override="RIMPHONY_CHECK_ORACLE_DIST"))

    s, gamma, n, cos_th, sin_th = pos("s gamma n cos_th sin_th")
    Jn, Jp = sp.symbols("Jn Jp")
    geo = {"s": s, "gamma": gamma, "n": n, "cos_th": cos_th, "sin_th": sin_th, "cos_observer_angle": cos_th, "sin_observer_angle": sin_th}

    # ================= gamma_integrand: kinematics (symphony.rs:398-442) =================
    ref_body = fn_body(sym_rs, "fn gamma_integrand(&mut self, gamma: f64, n: f64) -> f64")
    bess = {"leung_bessel": None}
    ref_body_k = ref_body.replace("leung_bessel::Jn(n, z)", "Jn").replace("leung_bessel::Jn_prime(n, z)", "Jp")

    def split_branch(body, lang):
        """(text before the if, naive branch, stabilised branch, text after) of the gamma_sin_xi computation"""
        i = body.index("if (beta < 0.1)") if lang == "c" else body.index("if beta < 0.1")
        j = body.index("{", i)
        naive = fn_body(body[i:], "if")
        k = body.index("else", j)
        stab = fn_body(body[k:], "else")
        after = body[body.index(stab, k) + len(stab):]
        after = after[after.index("}") + 1:]
        return body[:i], naive, stab, after

    pre, naive, stab, after = split_branch(ref_body_k, "rust")
    pre = pre[:pre.rindex("let gamma_sin_xi")]
    for branch, blk in (("beta < 0.1", "let gamma_sin_xi = " + naive.strip() + ";"),
                        ("beta >= 0.1", re.sub(r"\n\s*\(r \*", "\n let gamma_sin_xi = (r *", stab.strip()) + ";")):
        ref = run(pre + blk + after, base_env(dict(geo, Jn=Jn, Jp=Jp)), "rust")
        for label, text, head, lang in (("dev_symphony.h", dev_sym, "RIM_DEV GiShared gamma_integrand_shared(", "c"),
                                        ("oracle/rimo_symphony.c", ora_sym, "static double gamma_integrand(sym_state *st, double gamma, double n)", "c")):
            body = fn_body(text, head)
            body = body.replace("rimo_bessel_j(n, z)", "Jn").replace("rimo_bessel_dj(n, z)", "Jp")
            body = re.sub(r"const double s = st->s;|const double cos_th = st->cos_observer_angle, sin_th = st->sin_observer_angle;|"
                          r"const double s = pt\.s, n = so\.n;|const double n = so\.n;", "", body)
            body = re.sub(r"sym_bessel_pair<PREC>\(so, z, jn, djn\);", "jn = Jn; djn = Jp;", body)
            p2, nv, sb, af = split_branch(body, "c")
            p2 = re.sub(r"double gamma_sin_xi;\s*$", "", p2.rstrip()) + "\n"
            ours = run(p2 + (nv if branch == "beta < 0.1" else sb) + af, base_env(dict(geo, Jn=Jn, Jp=Jp, jn=Jn, djn=Jp)), lang)
            for q in ("beta", "cos_xi", "sin_xi", "m", "big_n", "gamma_sin_xi", "z", "mj", "njp"):
                qo = q
                if label == "dev_symphony.h" and q in ("mj", "njp"):
                    # GiShared fields: `sh.mj = m * jn;`
                    pass
                same("gamma_integrand %s (%s)" % (q, branch), ours.get(qo), ref.get(q), "symphony.rs:406-442 vs " + label)

    # ================= polarisation and distribution terms (symphony.rs:444-465) =================
    mj, njp, dfdg, dfdcx, beta, cos_xi = sp.symbols("mj njp dfdg dfdcx beta cos_xi")
    arms = dict(re.findall(r"Stokes::(\w) => ([^,]+),", between(ref_body, "let pol_term = match self.stokes", "};")))
    pol_dev = fn_body(dev_sym, "RIM_DEV double gamma_integrand_pol_term(")
    pol_ora = between(ora_sym, "switch (st->stokes) {", "double f_term;")
    for k, (stokes, c_case) in enumerate((("I", "RIMO_STOKES_I"), ("Q", "RIMO_STOKES_Q"), ("V", "default"))):
        ref = eval(exact(arms[stokes]), {"__builtins__": {}}, {"R": sp.Rational, "mj": mj, "njp": njp})
        dev_ret = re.findall(r"return ([^;]+);", pol_dev)[k]
        same("pol_term Stokes %s" % stokes, eval(exact(dev_ret), {"__builtins__": {}}, {"R": sp.Rational, "mj": mj, "njp": njp}), ref,
             "symphony.rs:444-448 vs dev_symphony.h")
        ora_rhs = re.search(r"%s:\s*pol_term = ([^;]+);" % ("case " + c_case if c_case != "default" else "default"), pol_ora).group(1)
        same("pol_term Stokes %s" % stokes, eval(exact(ora_rhs), {"__builtins__": {}}, {"R": sp.Rational, "mj": mj, "njp": njp}), ref,
             "symphony.rs:444-448 vs oracle/rimo_symphony.c")
    fenv = base_env({"beta": beta, "cos_xi": cos_xi, "gamma": gamma, "cos_th": cos_th, "cos_observer_angle": cos_th, "dfdg": dfdg, "dfdcx": dfdcx})
    absorb = between(ref_body, "Coefficient::Absorption => {", "Coefficient::Faraday")
    absorb = absorb[absorb.index("{") + 1:]
    refa = run(absorb.replace("dfdg + dfdcx_factor * dfdcx", "let f_term = dfdg + dfdcx_factor * dfdcx;"), fenv, "rust")
    fdev = fn_body(dev_sym, "RIM_DEV double gamma_integrand_f_term(")
    aniso = fdev[fdev.rindex("const double dfdcx_factor"):]
    oa = run(aniso.replace("return dfdg + dfdcx_factor * dfdcx", "f_term = dfdg + dfdcx_factor * dfdcx"), fenv, "c")
    same("absorption f_term (anisotropic distributions)", oa.get("f_term"), refa.get("f_term"), "symphony.rs:455-463 vs dev_symphony.h")
    iso = re.search(r"return (dfdg \+ \(\(beta \* cos_th - cos_xi\) \* dfdcx\) \* \(gamma - 1\.\));", fdev).group(1)
    iso_v = eval(exact(iso), {"__builtins__": {}}, dict(fenv, R=sp.Rational))
    same("absorption f_term (isotropic: dfdcx = 0, the division-free form)", iso_v.subs(dfdcx, 0), refa.get("f_term").subs(dfdcx, 0),
         "symphony.rs:455-463 vs dev_symphony.h")
    oo = between(ora_sym, "rimo_calc_f_derivatives(st->d, gamma, cos_xi, &dfdg, &dfdcx);", "if (st->c)")
    oo = run(oo.replace("f_term = dfdg", "f_term = dfdg"), fenv, "c")
    same("absorption f_term", oo.get("f_term"), refa.get("f_term"), "symphony.rs:455-463 vs oracle/rimo_symphony.c")
    pol_term, f_term = sp.symbols("pol_term f_term")
    final_ref = eval(exact(re.search(r"let r = (gamma \* gamma[^;]+);", ref_body).group(1)), {"__builtins__": {}}, {"R": sp.Rational, "gamma": gamma, "pol_term": pol_term, "f_term": f_term})
    final_dev = eval(exact(re.search(r"return (gamma \* gamma \* pol_term \* f_term);", dev_sym).group(1)), {"__builtins__": {}}, {"R": sp.Rational, "gamma": gamma, "pol_term": pol_term, "f_term": f_term})
    same("integrand = gamma^2 pol_term f_term", final_dev, final_ref, "symphony.rs:465 vs dev_symphony.h")

    # ================= gamma limits (symphony.rs:312-363) =================
    gi = fn_body(sym_rs, "fn gamma_integral(&mut self, workspace: &mut gsl::IntegrationWorkspace, n: f64) -> f64")
    gi_pre = gi[:gi.index("let rel_width")]
    gi_rw = re.search(r"\(-0\.27 \* n\.ln\(\) - 0\.1\)\.exp\(\)", gi).group(0)
    gi_post = between(gi, "let gamma_minus_high", "let (gamma0, gamma1)")
    for label, rw in (("s < 1e6: rel_width = 1", "1."), ("s >= 1e6", gi_rw)):
        ref = run(gi_pre + "let rel_width = %s;" % rw + gi_post, base_env(geo), "rust")
        dev = fn_body(wave_sym, "__device__ __forceinline__ GammaLimits gamma_limits(")
        dev = dev.replace("const double s = pt.s;", "").replace("(s < 1e6) ? 1. : rim_exp(-0.27 * rim_log(n) - 0.1)",
                                                                "1." if rw == "1." else "rim_exp(-0.27 * rim_log(n) - 0.1)")
        ours = run(dev, base_env(dict(geo, acos_th=None)), "c")
        ora = between(ora_sym, "static double gamma_integral(sym_state *st, double n)", "double gamma0, gamma1;")
        ora = ora.replace("const double s = st->s;", "").replace("(s < 1e6) ? 1. : m_exp(-0.27 * m_log(n) - 0.1)",
                                                                  "1." if rw == "1." else "m_exp(-0.27 * m_log(n) - 0.1)")
        oo = run(ora, base_env(geo), "c")
        for q in ("gamma_minus", "gamma_plus", "gamma_peak", "gamma_minus_high", "gamma_plus_high"):
            same("%s (%s)" % (q, label), ours.get(q), ref.get(q), "symphony.rs:315-343 vs symphony_wave.h")
            same("%s (%s)" % (q, label), oo.get(q), ref.get(q), "symphony.rs:315-343 vs oracle/rimo_symphony.c")
    lobes = re.findall(r"StokesVSwitch::(\w+) => \((\w+), (\w+)\)", gi)
    dev_gl = fn_body(wave_sym, "__device__ __forceinline__ GammaLimits gamma_limits(")
    ours_lobes = [("PositiveLobe",) + re.search(r"if \(!negative_lobe\) \{ L\.g0 = (\w+); L\.g1 = (\w+); \}", dev_gl).groups(),
                  ("NegativeLobe",) + re.search(r"else \{ L\.g0 = (\w+); L\.g1 = (\w+); \}\s*\} else", dev_gl).groups()]
    ok = sorted(lobes) == sorted(ours_lobes) and re.search(r"else \{\s*L\.g0 = gamma_minus_high;\s*L\.g1 = gamma_plus_high;", dev_gl) is not None
    rows.append("%-74s %s" % ("lobe limits (V: peak..plus_high / minus_high..peak; else minus_high..plus_high)  [symphony.rs:355-363 vs symphony_wave.h]", "identical" if ok else "DIFFERENT"))
    print(rows[-1])
    if not ok:
        failures.append("lobe limits")

    # ================= prefactors (symphony.rs:173-183) and cgs layer (lib.rs:55-67, 163-173) =================
    consts = {k: sp.Rational(v) for k, v in re.findall(r"pub const (MASS_ELECTRON|SPEED_LIGHT|ELECTRON_CHARGE): f64 = ([-\d.eE]+);", lib_rs)}
    consts["TWO_PI"] = 2 * sp.pi
    em = re.search(r"Coefficient::Emission => \{(.*?)\},", between(sym_rs, "let ans = ans * match self.coeff", "Coefficient::Faraday"), re.S).group(1)
    ab = re.search(r"Coefficient::Absorption => \{(.*?)\},", between(sym_rs, "let ans = ans * match self.coeff", "Coefficient::Faraday"), re.S).group(1)
    penv = base_env(dict(consts, cos_observer_angle=cos_th, cos_th=cos_th))
    ref_em = eval(exact(normalise(em, "rust").replace("\n", " ").strip()), {"__builtins__": {}}, dict(penv, R=sp.Rational))
    ref_ab = eval(exact(normalise(ab, "rust").replace("\n", " ").strip()), {"__builtins__": {}}, dict(penv, R=sp.Rational))
    for label, text, head in (("symphony_wave.h", wave_sym, "__device__ __forceinline__ double sym_result("),
                              ("oracle/rimo_symphony.c", ora_sym, "const double tpe = TWO_PI * ELECTRON_CHARGE;")):
        blk = fn_body(text, head) if label == "symphony_wave.h" else text[text.index(head):text.index("done:")]
        cdef = {}
        for k in ("MASS_ELECTRON", "SPEED_LIGHT", "ELECTRON_CHARGE"):
            v = re.search(r"#define\s+(?:RIM_)?%s\s+([-\d.eE]+)" % k, text).group(1)
            cdef[k] = cdef["RIM_" + k] = sp.Rational(v)
            same("constant %s = %s" % (k, v), cdef[k], consts[k], "lib.rs:55-67 vs " + label)
        cdef["TWO_PI"] = cdef["RIM_TWO_PI"] = 2 * sp.pi
        ans = sp.Symbol("ans")
        e = run(blk.replace("rim_fabs(pt.cos_th)", "cos_th").replace("m_fabs(st.cos_observer_angle)", "cos_th"),
                base_env(dict(cdef, cos_th=cos_th, ans=ans)), "c", only=("tpe", "acos_th"))
        e["ans"] = ans
        fac = re.findall(r"ans = ans \* (\(.*?\));", strip_comments(blk), re.S)
        vals = [eval(exact(normalise(x, "c")), {"__builtins__": {}}, dict(e, R=sp.Rational)) for x in fac[:2]]
        same("emission prefactor (2 pi e)^2 / (c |cos theta|)", vals[0], ref_em, "symphony.rs:174-176 vs " + label)
        same("absorption prefactor -(2 pi e)^2 / (2 m_e c |cos theta|)", vals[1], ref_ab, "symphony.rs:178-181 vs " + label)
    # compute_cgs: nu_c = e B / (2 pi m_e c); val(s = nu / nu_c) x {n_e nu, n_e / nu, n_e / nu}
    cg = fn_body(lib_rs, "fn compute_cgs(")
    nu, B, n_e, val = pos("nu b n_e val")
    ref_cg = run(cg, base_env(dict(consts, nu=nu, b=B, n_e=n_e)), "rust", only=("nu_c",))
    ora_cg = fn_body(ora_sym, "double rimo_compute_cgs(")
    our_cg = run(ora_cg, base_env(dict(consts, nu=nu, b=B, n_e=n_e)), "c", only=("nu_c",))
    same("cyclotron frequency nu_c = e B / (2 pi m_e c)", our_cg.get("nu_c"), ref_cg.get("nu_c"), "lib.rs:165 vs oracle/rimo_symphony.c")
    arms_cg = dict(re.findall(r"Coefficient::(\w+) => ([^,]+),", cg))
    cenv = {"R": sp.Rational, "val": val, "n_e": n_e, "nu": nu}
    o_em = re.search(r"if \(coeff == RIMO_EMISSION\) return ([^;]+);", ora_cg).group(1)
    o_other = re.findall(r"return ([^;]+);", ora_cg)[-1]
    same("cgs scaling, emission: val n_e nu", eval(exact(o_em), {"__builtins__": {}}, cenv), eval(exact(arms_cg["Emission"]), {"__builtins__": {}}, cenv), "lib.rs:168-172 vs oracle/rimo_symphony.c")
    for arm in ("Absorption", "Faraday"):
        same("cgs scaling, %s: val n_e / nu" % arm.lower(), eval(exact(o_other), {"__builtins__": {}}, cenv), eval(exact(arms_cg[arm]), {"__builtins__": {}}, cenv), "lib.rs:168-172 vs oracle/rimo_symphony.c")
    api = read(ROOT, "rimphony_amd", "api.py")
    pyc = {k: sp.Rational(v) for k, v in re.findall(r"^(MASS_ELECTRON|SPEED_LIGHT|ELECTRON_CHARGE) = ([-\d.eE]+)", api, re.M)}
    for k in ("MASS_ELECTRON", "SPEED_LIGHT", "ELECTRON_CHARGE"):
        same("constant %s" % k, pyc.get(k), consts[k], "lib.rs:55-67 vs rimphony_amd/api.py")

    # ================= distributions (calc_f, calc_f_derivatives) =================
    norm, p, gcut_inv, k_, kappa, width, ikw, nit = pos("norm p inv_gamma_cutoff k kappa width inv_kappa_width neg_inverse_t_abs")
    cx = sp.Symbol("cos_xi", real=True)
    denv = base_env({"gamma": gamma, "cos_xi": cx, "_cos_xi": cx, "norm": norm, "p": p, "inv_gamma_cutoff": gcut_inv, "k": k_,
                     "kappa": kappa, "width": width, "inv_kappa_width": ikw, "neg_inverse_t": -nit})
    dists = [("power_law.rs", "PowerLawDistribution", "DIST_POWER_LAW", "RIMO_POWER_LAW", {"par0": p}, "36-62"),
             ("thermal_juettner.rs", "ThermalJuettnerDistribution", "DIST_THERMAL_JUETTNER", "RIMO_THERMAL_JUETTNER", {}, "29-39"),
             ("pitchy_pl.rs", "PitchyPowerLawDistribution", "DIST_PITCHY_PL", "RIMO_PITCHY_PL", {"par0": p, "par1": k_}, "32-64"),
             ("pitchy_kappa.rs", "PitchyKappaDistribution", "DIST_PITCHY_KAPPA", "RIMO_PITCHY_KAPPA", {"par0": kappa, "par1": width, "par2": k_}, "38-62")]
    dev_f = fn_body(dev_sym, "RIM_DEV double calc_f(const DistParams &d, double gamma, double cos_xi)")
    dev_d = fn_body(dev_sym, "RIM_DEV void calc_f_derivatives(const DistParams &d, double gamma, double cos_xi, double &dfdg, double &dfdcx)")
    kap = fn_body(dev_sym, "RIM_DEV double kappa_gamma_term(const DistParams &d, double gamma)")
    kap_e = run(kap.replace("return rim_pow(base, y) * RimMath<PREC>::exp(-gamma * d.inv_gamma_cutoff)", "kgt = rim_pow(base, y) * exp(-gamma * inv_gamma_cutoff)"),
                dict(denv, par0=kappa, par1=width), "c")
    ora_f = fn_body(ora_dist, "double rimo_calc_f(const rimo_dist *d, double gamma, double cos_xi)")
    ora_d = fn_body(ora_dist, "void rimo_calc_f_derivatives(const rimo_dist *d, double gamma, double cos_xi, double *dfdg, double *dfdcx)")
    ora_kap = fn_body(ora_dist, "static double kappa_gamma_term(const rimo_dist *d, double gamma)")
    ora_kap_e = run(re.sub(r"#ifndef RIMO_LIBM.*?#endif", "", ora_kap, flags=re.S).replace("return m_pow(base, y) * m_exp(e)", "kgt = m_pow(base, y) * m_exp(e)"),
                    dict(denv, par0=kappa, par1=width), "c")

    def branch_of(text, tag, nxt):
        """the block of `if (KIND == tag) {` / `case tag: {` up to the next one"""
        i = text.index(tag)
        j = text.index(nxt, i) if nxt and nxt in text[i:] else len(text)
        return text[i:j]

    tags_dev = [d[2] for d in dists] + [None]
    tags_ora = [d[3] for d in dists] + [None]
    for idx, (fname, struct, tdev, tora, pars, lines) in enumerate(dists):
        rs = read(REF, "src", fname)
        rf = fn_body(rs, "fn calc_f(&self, gamma: f64, %scos_xi: f64) -> f64" % ("_" if "_cos_xi" in rs else ""))
        rd = fn_body(rs, "fn calc_f_derivatives(&self, gamma: f64, %scos_xi: f64) -> (f64, f64)" % ("_" if "_cos_xi" in rs else ""))
        # the value of calc_f is the last expression of the (else) block
        rf_stmts = re.sub(r"if gamma < self\.gamma_min \|\| gamma > self\.gamma_max \{[^}]*\}( else \{)?", "", rf)
        rf_stmts = rf_stmts.rstrip().rstrip("}").rstrip()
        last = rf_stmts[rf_stmts.rindex(";") + 1:] if ";" in rf_stmts else rf_stmts
        head = rf_stmts[:rf_stmts.rindex(";") + 1] if ";" in rf_stmts else ""
        ref = run(head + " let f_value = " + last.strip() + ";", denv, "rust")
        rd_stmts = re.sub(r"if gamma < self\.gamma_min \|\| gamma > self\.gamma_max \{[^}]*\}", "", rd)
        refd = run(rd_stmts, denv, "rust")
        e0 = dict(denv, kappa_gamma_term=lambda *a: kap_e["kgt"], d=None, **pars)
        bdev = branch_of(dev_f, tdev, tags_dev[idx + 1] if idx < 2 else "} else {") if idx < 3 else dev_f[dev_f.rindex("} else {"):]
        bdev = re.sub(r"if \(gamma < d\.par\[\d\] \|\| gamma > d\.par\[\d\]\) return 0\.;", "", bdev)
        bdev = re.sub(r"return ([^;]+);", r"f_value = \1;", bdev)
        ours = run(bdev, e0, "c")
        same("%s::calc_f" % struct, ours.get("f_value"), ref.get("f_value"), "%s:%s vs dev_symphony.h" % (fname, lines))
        ddev = branch_of(dev_d, tdev, tags_dev[idx + 1] if idx < 2 else "} else {") if idx < 3 else dev_d[dev_d.rindex("} else {"):]
        ddev = re.sub(r"if \(gamma < d\.par\[\d\] \|\| gamma > d\.par\[\d\]\) \{[^}]*\}", "", ddev)
        oursd = run(ddev, e0, "c")
        same("%s::calc_f_derivatives dfdg" % struct, oursd.get("dfdg"), refd.get("dfdg"), "%s:%s vs dev_symphony.h" % (fname, lines))
        same("%s::calc_f_derivatives dfdcx" % struct, oursd.get("dfdcx"), refd.get("dfdcx"), "%s:%s vs dev_symphony.h" % (fname, lines))
        e1 = dict(denv, kappa_gamma_term=lambda *a: ora_kap_e["kgt"], d=None, **pars)
        e1.update({"p": p, "k": k_, "kappa": kappa, "width": width})
        bo = branch_of(ora_f, "case " + tora, ("case " + tags_ora[idx + 1]) if tags_ora[idx + 1] else "return RIM_NAN")
        bo = re.sub(r"#ifdef RIMO_LIBM(.*?)#else.*?#endif", r"\1", bo, flags=re.S)          # the literal flavour's form
        bo = re.sub(r"if \(gamma < d->par\[\d\] \|\| gamma > d->par\[\d\]\)\s*return 0\.;", "", bo)
        bo = re.sub(r"return ([^;]+);", r"f_value = \1;", bo)
        oo = run(bo, e1, "c")
        same("%s::calc_f" % struct, oo.get("f_value"), ref.get("f_value"), "%s:%s vs oracle/rimo_dist.c" % (fname, lines))
        bd = branch_of(ora_d, "case " + tora, ("case " + tags_ora[idx + 1]) if tags_ora[idx + 1] else "\n}\n")
        bd = re.sub(r"#ifdef RIMO_LIBM(.*?)#else.*?#endif", r"\1", bd, flags=re.S)
        bd = bd[:bd.index("return;") + 7] if "return;" in bd[20:] and idx == 3 else bd
        bd = re.sub(r"if \(gamma < d->par\[\d\] \|\| gamma > d->par\[\d\]\) \{[^}]*\}", "", bd)
        bd = bd.replace("*dfdg", "dfdg").replace("*dfdcx", "dfdcx")
        ood = run(bd, e1, "c")
        same("%s::calc_f_derivatives dfdg" % struct, ood.get("dfdg"), refd.get("dfdg"), "%s:%s vs oracle/rimo_dist.c" % (fname, lines))
        same("%s::calc_f_derivatives dfdcx" % struct, ood.get("dfdcx"), refd.get("dfdcx"), "%s:%s vs oracle/rimo_dist.c" % (fname, lines))

    # ================= normalisation integrands and norm = 1 / (4 pi [pa] int) =================
    g = pos("g")
    hip = read(ROOT, "rimphony_amd", "csrc", "rimphony_hip.hip")
    nenv = base_env({"g": g, "p": p, "par0": p, "inv_gamma_cutoff": gcut_inv, "inv_kappa_width": ikw, "kappa": kappa, "TWO_PI": 2 * sp.pi,
                     "RIM_TWO_PI": 2 * sp.pi})
    def closure(src, start):
        i = src.index(start) + len(start)
        j, depth = i, 0
        while True:
            if src[j] in "({":
                depth += 1
            elif src[j] in ")}":
                if depth == 0:
                    break
                depth -= 1
            elif src[j] == "," and depth == 0:
                break
            j += 1
        return src[i:j]
    pl_rs = read(REF, "src", "power_law.rs")
    pk_rs = read(REF, "src", "pitchy_kappa.rs")
    ref_pl = eval(exact(normalise(closure(pl_rs, "ws.qag(|g| "), "rust")), {"__builtins__": {}}, dict(nenv, R=sp.Rational))
    ref_pp = eval(exact(normalise(closure(read(REF, "src", "pitchy_pl.rs"), "ws.qag(|g| "), "rust")), {"__builtins__": {}}, dict(nenv, R=sp.Rational))
    fg = between(pk_rs, "let fg = |g: f64| {", "};")
    ref_pk = eval(exact(normalise(fg[fg.index("{") + 1:], "rust").replace("\n", " ")), {"__builtins__": {}}, dict(nenv, R=sp.Rational))
    dev_n = fn_body(hip, "__device__ inline double norm_integrand(const DistParams &d, double g)")
    dev_rets = [normalise(x, "c").replace("\n", " ") for x in re.findall(r"return ([^;]+);", dev_n)]
    ora_pl = normalise(re.search(r"return (m_pow\(g, -p\)[^;]+);", ora_dist).group(1), "c")
    ora_pk = normalise(re.search(r"return (g \* m_sqrt\(g \* g - 1\.\)[^;]+);", ora_dist, re.S).group(1), "c").replace("\n", " ")
    same("normalisation integrand gamma^-p exp(-gamma / gamma_c)", eval(exact(dev_rets[0]), {"__builtins__": {}}, dict(nenv, R=sp.Rational)), ref_pl, "power_law.rs:95 vs rimphony_hip.hip")
    same("normalisation integrand gamma^-p exp(-gamma / gamma_c)", eval(exact(dev_rets[0]), {"__builtins__": {}}, dict(nenv, R=sp.Rational)), ref_pp, "pitchy_pl.rs:102 vs rimphony_hip.hip")
    same("normalisation integrand gamma^-p exp(-gamma / gamma_c)", eval(exact(ora_pl), {"__builtins__": {}}, dict(nenv, R=sp.Rational)), ref_pl, "power_law.rs:95 vs oracle/rimo_dist.c")
    same("normalisation integrand of the kappa distribution", eval(exact(dev_rets[1]), {"__builtins__": {}}, dict(nenv, R=sp.Rational, par0=kappa)), ref_pk, "pitchy_kappa.rs:100-104 vs rimphony_hip.hip")
    same("normalisation integrand of the kappa distribution", eval(exact(ora_pk), {"__builtins__": {}}, dict(nenv, R=sp.Rational)), ref_pk, "pitchy_kappa.rs:100-104 vs oracle/rimo_dist.c")
    # thermal: the reference integrates gamma sqrt(gamma^2 - 1) exp(-gamma / T) over [1, inf) (QAGIU); kernels and oracle
    # integrate the same function substituted gamma = 1 + u^2 (documented deviation 1 of DESIGN.md section 2)
    tj_rs = read(REF, "src", "thermal_juettner.rs")
    u = pos("u")
    ref_tj = eval(exact(normalise(closure(tj_rs, "ws.qagiu(|g| "), "rust")) if "ws.qagiu(|g| " in tj_rs else "None", {"__builtins__": {}}, dict(nenv, R=sp.Rational, neg_inverse_t=-nit))
    if ref_tj is not None:
        sub = ref_tj.subs(g, 1 + u ** 2) * 2 * u
        dev_tj = eval(exact(dev_rets[2].replace("gg", "GG")), {"__builtins__": {}}, dict(nenv, R=sp.Rational, u=u, u2=u * u, GG=1 + u * u, neg_inverse_t=-nit))
        same("thermal normalisation integrand under gamma = 1 + u^2", dev_tj, sub, "thermal_juettner.rs:58-62 vs rimphony_hip.hip")
    integral, pa = pos("integral pa_integral")
    for fname, pat, extra in (("power_law.rs", r"self\.norm = (1\. / \(2\. \* TWO_PI \* integral\));", {}),
                              ("pitchy_pl.rs", r"self\.norm = (1\. / \(2\. \* TWO_PI \* pa_integral \* gamma_integral\));", {}),
                              ("pitchy_kappa.rs", r"self\.norm = (1\. / \(2\. \* TWO_PI \* pa_integral \* gamma_integral\));", {})):
        refn = eval(exact(re.search(pat, read(REF, "src", fname)).group(1)), {"__builtins__": {}}, {"R": sp.Rational, "TWO_PI": 2 * sp.pi, "integral": integral, "pa_integral": pa, "gamma_integral": integral})
        with_pa = "pa_integral" in pat
        o = re.search(r"d->norm = (1\. / \(2\. \* TWO_PI \* %sintegral\));" % ("pa_integral \\* " if with_pa else ""), ora_dist).group(1)
        same("norm = 1 / (4 pi %sintegral)" % ("pa_integral " if with_pa else ""), eval(exact(o), {"__builtins__": {}}, {"R": sp.Rational, "TWO_PI": 2 * sp.pi, "integral": integral, "pa_integral": pa}), refn, "%s vs oracle/rimo_dist.c" % fname)
        k2 = re.search(r"v = (1\. / \(2\. \* RIM_TWO_PI \* %sq\.result\));" % ("pa \\* " if with_pa else ""), hip).group(1)
        same("norm = 1 / (4 pi %sintegral)" % ("pa_integral " if with_pa else ""), eval(exact(k2.replace("q.result", "integral")), {"__builtins__": {}}, {"R": sp.Rational, "RIM_TWO_PI": 2 * sp.pi, "integral": integral, "pa": pa}), refn, "%s vs rimphony_hip.hip norm_kernel" % fname)

    # ================= Heyvaerts: coordinates, chain rule, inner limits, scaling =================
    sig, po, s0, s0sq = pos("sigma pomega sigma0 sigma0_sq")
    # (round 4: the kernels multiply by dinv = 1 / (sigma0 sin_th), formed once per coefficient by hey_point_derive)
    assert "pt.dinv = 1. / (pt.sigma0 * pt.sin_th);" in dev_hey
    henv = base_env({"sigma": sig, "pomega": po, "sigma0": s0, "sigma0_sq": s0sq, "cos_th": cos_th, "sin_th": sin_th, "dinv": 1 / (s0 * sin_th),
                     "cos_observer_angle": cos_th, "sin_observer_angle": sin_th,
                     "INVERSE_SQRT_3": sp.Symbol("ISQ3"), "RIM_INVERSE_SQRT_3": sp.Symbol("ISQ3"),
                     "THREE_TWO_THIRDS": sp.Symbol("T23"), "RIM_THREE_TWO_THIRDS": sp.Symbol("T23")})
    rfc = fn_body(hey_rs, "fn fill_coord_vars(&mut self, sigma: f64, pomega: f64)")
    ref = run(rfc, henv, "rust")
    for label, text, head in (("dev_heyvaerts.h", dev_hey, "RIM_DEV HeyCoord fill_coord_vars("), ("oracle/rimo_heyvaerts.c", ora_hey, "static void fill_coord_vars(")):
        ours = run(fn_body(text, head).replace("HeyCoord c;", ""), henv, "c")
        for q in ("x", "gamma", "mu"):
            same("fill_coord_vars %s" % q, ours.get(q), ref.get(q), "heyvaerts.rs:194-201 vs " + label)
    dfdcxi = sp.Symbol("dfdcxi")
    hg, hmu = pos("gamma_h mu_h")
    denv2 = dict(henv, dfdg=dfdg, dfdcxi=dfdcxi, gamma=hg, mu=hmu)
    rdf = fn_body(hey_rs, "fn dfdsigma(&self) -> f64")
    mu_blk = fn_body(rdf[rdf.index("let mu_term"):], "else")
    ref = run("let g_term = dfdg / (self.sigma0 * self.sin_observer_angle);" + re.sub(r"dcxi_dsigma \* dfdcxi\s*$", "let mu_term = dcxi_dsigma * dfdcxi;", mu_blk.strip()), denv2, "rust")
    for label, text, head in (("dev_heyvaerts.h", dev_hey, "RIM_DEV double dfdsigma("), ("oracle/rimo_heyvaerts.c", ora_hey, "static double dfdsigma(")):
        body = fn_body(text, head)
        g_stmt = re.search(r"const double g_term = [^;]+;", body).group(0)
        blk = fn_body(body[body.index("if (dfdcxi == 0.)"):], "else")
        ours = run(g_stmt + blk, denv2, "c")
        same("dfdsigma g_term", ours.get("g_term"), ref.get("g_term"), "heyvaerts.rs:472-476 vs " + label)
        same("dfdsigma mu_term (chain rule through mu)", ours.get("mu_term"), ref.get("mu_term"), "heyvaerts.rs:478-490 vs " + label)
    # limits of the inner integrals
    u_l = pos("u_l")
    rn = fn_body(hey_rs, "fn nr_outer_integrand(")
    rq = fn_body(hey_rs, "fn qr_outer_integrand(")
    ref_n = run(rn[:rn.index("if sigma_max <= sigma_min")], dict(henv, pomega=u_l), "rust")
    ref_q = run(rq[:rq.index("match self.stokes")], dict(henv, sigma=u_l), "rust")
    hev = fn_body(wave_hey, "__device__ __forceinline__ void hey_eval_pair(") if "hey_eval_pair(" in wave_hey else wave_hey
    i0 = wave_hey.index("const double sigma_min = rim_sqrt(u_l * u_l + pt.sigma0_sq);")
    ours_n = run(wave_hey[i0:wave_hey.index("empty_l", i0)], dict(henv, u_l=u_l), "c")
    j0 = wave_hey.index("const double pomega_max_phys")
    ours_q = run(wave_hey[j0:wave_hey.index("lo_l = -pomega_max", j0)], dict(henv, u_l=u_l), "c")
    on = fn_body(ora_hey, "static double nr_outer_integrand(") if "static double nr_outer_integrand(" in ora_hey else ora_hey
    oi = ora_hey.index("const double sigma_min = m_sqrt(pomega * pomega + st->sigma0_sq);")
    oo_n = run(ora_hey[oi:ora_hey.index("if (sigma_max <= sigma_min)", oi)], dict(henv, pomega=u_l), "c")
    oj = ora_hey.index("const double pomega_max_phys = m_sqrt(")
    oo_q = run(ora_hey[oj:ora_hey.index("return inner_qag(st, qr_inner_cb", oj)], dict(henv, sigma=u_l), "c")
    for q, r_, o1, o2, ln in (("sigma_min", ref_n, ours_n, oo_n, "214"), ("sigma_max", ref_n, ours_n, oo_n, "215"),
                              ("pomega_max_phys", ref_q, ours_q, oo_q, "263"), ("pomega_max_qr", ref_q, ours_q, oo_q, "264"),
                              ("pomega_max", ref_q, ours_q, oo_q, "265")):
        same("inner limit %s" % q, o1.get(q), r_.get(q), "heyvaerts.rs:%s vs heyvaerts_wave.h" % ln)
        same("inner limit %s" % q, o2.get(q), r_.get(q), "heyvaerts.rs:%s vs oracle/rimo_heyvaerts.c" % ln)
    # final scaling 2 e^2 (nr + qr) / (m_e (s sin theta)^2)
    nr_val, qr_val = sp.symbols("nr_val qr_val")
    fin = re.search(r"2\. \* ELECTRON_CHARGE\.powi\(2\) \* \(nr_val \+ qr_val\) /\s*\(MASS_ELECTRON \* \(self\.s \* self\.sin_observer_angle\)\.powi\(2\)\)", hey_rs).group(0)
    ref_fin = eval(exact(normalise(fin, "rust").replace("\n", " ")), {"__builtins__": {}}, dict(base_env(dict(consts, s=s, sin_observer_angle=sin_th, nr_val=nr_val, qr_val=qr_val)), R=sp.Rational))
    for label, text in (("heyvaerts_wave.h", wave_hey), ("oracle/rimo_heyvaerts.c", ora_hey)):
        m = re.search(r"=\s*(2\. \* \((?:RIM_)?ELECTRON_CHARGE[^;]*);", text)
        expr = normalise(m.group(1), "c").replace("\n", " ")
        cenv = dict(base_env(dict(consts, s=s, sin_th=sin_th, sin_observer_angle=sin_th, nr_val=nr_val, qr_val=qr_val, sigma0=s * sin_th, sigma0_sq=(s * sin_th) ** 2, ssin=s * sin_th)), R=sp.Rational)
        for k in ("MASS_ELECTRON", "ELECTRON_CHARGE"):
            cenv["RIM_" + k] = consts[k]
        same("Faraday final scaling 2 e^2 (nr + qr) / (m_e (s sin theta)^2)", eval(exact(expr), {"__builtins__": {}}, cenv), ref_fin, "heyvaerts.rs:189-190 vs " + label)

    print("\n%d quantities compared, %d DIFFERENT" % (len(rows), len(failures)))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
