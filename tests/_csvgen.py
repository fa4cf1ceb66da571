"""Random tables and their CSV text in the dialect the reference's harness reads (Table::from_csv,
reference src/build_table.cpp:231: separator ',', escape '\\', quotes '"', no header): a small model
used by the ingest tests — rows as Python values (None = NULL, int, bytes), a writer that quotes
and escapes at random wherever the dialect allows it, and an independent row-level evaluator of the
filter programs."""
import numpy as np

INT32, INT64, FP64, VARCHAR = 0, 1, 2, 3
_SPECIAL = b',"\n\r'


class FText(float):
    """an FP64 cell: the double a correctly rounding reader makes of `text` (Python's float() is
    one), together with that text — the CSV writer emits it as it stands"""

    def __new__(cls, text: str):
        self = super().__new__(cls, text)
        self.text = text
        return self


def random_fp64_text(rng) -> str:
    """_random_fp64_text() without the texts whose value no double represents (overflow, or non-zero
    text that rounds to zero): those are errors in the reference and tested on their own"""
    s = _random_fp64_text(rng)
    v = float(s)
    sig = s.lower().split("e")[0]
    if v in (float("inf"), float("-inf")) or (v == 0 and any(ch in "123456789" for ch in sig)):
        return "0.25"
    return s


def _random_fp64_text(rng) -> str:
    """decimal text of the plain grammar std::from_chars(double) reads: sign, digits, point,
    exponent; short and long significands, values across the whole exponent range (subnormals
    included), texts that sit next to a rounding boundary"""
    k = int(rng.integers(0, 9))
    bits = int(rng.integers(0, 2**63, dtype=np.uint64)) | (int(rng.integers(0, 2)) << 63)
    d = float(np.array([bits], dtype=np.uint64).view(np.float64)[0])
    if not np.isfinite(d):
        d = 1.5
    if k == 0:
        return repr(d)  # shortest text that reads back as d
    if k == 1:
        return "%.*e" % (int(rng.integers(0, 24)), d)
    if k == 2:
        return "%d.%0*d" % (int(rng.integers(-10**6, 10**6)), int(rng.integers(1, 10)), int(rng.integers(0, 10**9)))
    if k == 3:
        digs = "".join(str(int(x)) for x in rng.integers(0, 10, int(rng.integers(1, 32))))
        at = int(rng.integers(0, len(digs) + 1))
        s = digs[:at] + ("." if rng.random() < 0.6 else "") + digs[at:]
        if rng.random() < 0.5:
            s += "eE"[int(rng.integers(0, 2))] + ["", "+", "-"][int(rng.integers(0, 3))] + str(int(rng.integers(0, 330)))
        return ("-" if rng.random() < 0.3 else "") + s
    if k == 4:  # half-way between two doubles, cut off after 17..40 digits
        from decimal import Decimal, getcontext

        getcontext().prec = 60
        a = abs(d) if abs(d) < 1e300 else 1.0
        mid = (Decimal(a) + Decimal(float(np.nextafter(a, np.inf)))) / 2
        return ("%.*e" % (int(rng.integers(16, 40)), mid)) if mid else "0"
    if k == 5:  # subnormals
        sub = float(np.array([int(rng.integers(1, 2**52))], dtype=np.uint64).view(np.float64)[0])
        return "%.*e" % (int(rng.integers(0, 20)), sub)
    if k == 6:
        return str(int(rng.integers(-(2**63), 2**63 - 1)))
    if k == 7:
        return ["0", "-0", "0.0", "-0.0e5", ".5", "5.", "-.5", "1E5", "1e+5", "4.9e-324", "1.7976931348623157e308"][int(rng.integers(0, 11))]
    return "%.3f" % (float(rng.integers(-10**6, 10**6)) / 1000)


def random_rows(rng, n, types, null_p=0.1, long_p=0.0):
    cols = []
    for ty in types:
        nulls = rng.random(n) < null_p
        if ty == INT32:
            v = rng.integers(-(2**31), 2**31 - 1, n)
            small = rng.random(n) < 0.5
            v[small] = rng.integers(-50, 2050, int(small.sum()))
            cols.append([None if nulls[i] else int(v[i]) for i in range(n)])
        elif ty == INT64:
            v = rng.integers(-(2**63), 2**63 - 1, n)
            cols.append([None if nulls[i] else int(v[i]) for i in range(n)])
        elif ty == FP64:
            cols.append([None if nulls[i] else FText(random_fp64_text(rng)) for i in range(n)])
        else:
            alphabet = np.frombuffer(b"abcdefghij KLMNOP0123456789,\"\\\n\r;:'-_", dtype=np.uint8)
            lens = rng.integers(1, 40, n)
            lens[rng.random(n) < 0.02] = rng.integers(200, 1500, int((rng.random(n) < 0.02).sum()) or 1)[0]
            col = []
            for i in range(n):
                if nulls[i]:
                    col.append(None)
                    continue
                ln = int(lens[i])
                if long_p and rng.random() < long_p:
                    ln = int(rng.integers(8186, 30000))
                col.append(alphabet[rng.integers(0, len(alphabet), ln)].tobytes())
            cols.append(col)
    return [tuple(c[i] for c in cols) for i in range(n)]


def field_text(rng, v):
    if v is None:
        return b"" if rng.random() < 0.8 else b'""'  # an empty field is NULL, quoted or not
    if isinstance(v, FText):
        s = v.text.encode()
        return b'"' + s + b'"' if rng.random() < 0.1 else s
    if isinstance(v, (int, np.integer)):
        s = str(int(v)).encode()
        return b'"' + s + b'"' if rng.random() < 0.1 else s
    must = any(ch in _SPECIAL for ch in v)
    if must or rng.random() < 0.3:
        # inside quotes a backslash escapes '"' and itself; before any other character it stands
        # for itself, so it MAY be left alone there — but not at the end (it would take the quote)
        out = bytearray(b'"')
        for k, ch in enumerate(v):
            if ch == 0x22:
                out += b'\\"'
            elif ch == 0x5C:
                nxt = v[k + 1] if k + 1 < len(v) else 0x22
                out += b"\\\\" if nxt in (0x22, 0x5C) or rng.random() < 0.5 else b"\\"
            else:
                out.append(ch)
        out += b'"'
        return bytes(out)
    return v


def to_csv(rng, rows, crlf_p=0.2, final_newline=True):
    out = bytearray()
    for r, row in enumerate(rows):
        out += b",".join(field_text(rng, v) for v in row)
        last = r + 1 == len(rows)
        if last and not final_newline:
            break
        x = rng.random()
        out += b"\r\n" if x < crlf_p else (b"\r" if x < crlf_p * 1.2 else b"\n")
    return bytes(out)


def like_model(text: bytes, pattern: bytes) -> bool:
    """what the reference asks RE2 for (statement.h:118-161), through Python's re on well-formed
    UTF-8: '%' -> '.*', '_' -> '.', full match, '.' stops at a newline"""
    import re

    try:
        t, p = text.decode("utf-8"), pattern.decode("utf-8")
    except UnicodeDecodeError:
        return False  # ill-formed text matches nothing; an ill-formed pattern does not compile
    rx = "".join(".*" if ch == "%" else "." if ch == "_" else re.escape(ch) for ch in p)
    return re.fullmatch(rx, t) is not None


def eval_filter(prog, row, r):
    """row-level twin of the reference's bitmap arithmetic (statement.cpp:8-135,186-201)"""
    if not prog:
        return True
    st = []
    for term in prog:
        op = term[0]
        if op in ("AND", "OR"):
            b, a = st.pop(), st.pop()
            st.append((a and b) if op == "AND" else (a or b))
        elif op == "NOT":
            st.append(not st.pop())
        elif op == "BITMAP":
            st.append(bool((term[1][r >> 3] >> (r & 7)) & 1))
        elif op in ("LIKE", "NOT_LIKE"):
            x = row[term[1]]
            hit = x is not None and like_model(x, term[2])
            st.append(x is not None and (hit if op == "LIKE" else not hit))
        elif op == "IS_NULL":
            st.append(row[term[1]] is None)
        elif op == "IS_NOT_NULL":
            st.append(row[term[1]] is not None)
        else:
            x, y = row[term[1]], term[2]
            if x is None:
                st.append(False)
            else:
                st.append({"EQ": x == y, "NEQ": x != y, "LT": x < y, "GT": x > y, "LEQ": x <= y, "GEQ": x >= y}[op])
    assert len(st) == 1
    return st[0]


def random_filter(rng, rows, types, depth=3):
    n = len(rows)
    leaves = []
    for c, ty in enumerate(types):
        leaves.append(lambda c=c: ("IS_NULL", c))
        leaves.append(lambda c=c: ("IS_NOT_NULL", c))
        if ty == INT32:
            leaves.append(lambda c=c: (["EQ", "NEQ", "LT", "GT", "LEQ", "GEQ"][int(rng.integers(0, 6))], c, int(rng.integers(-100, 2100))))
        elif ty == INT64:
            leaves.append(lambda c=c: (["LT", "GT", "LEQ", "GEQ"][int(rng.integers(0, 4))], c, int(rng.integers(-(2**62), 2**62))))
        elif ty == FP64:
            def fcmp(c=c):
                pool = [float(row[c]) for row in rows[:50] if row[c] is not None] or [0.0]
                lit = pool[int(rng.integers(0, len(pool)))] if rng.random() < 0.6 else float(rng.normal()) * 10.0 ** int(rng.integers(-5, 6))
                return (["EQ", "NEQ", "LT", "GT", "LEQ", "GEQ"][int(rng.integers(0, 6))], c, float(lit))
            leaves.append(fcmp)
        else:
            # string comparisons run on the device (std::string order: unsigned bytes, then length)
            def strcmp(c=c):
                pool = [row[c] for row in rows[:50] if row[c] is not None] or [b"k"]
                lit = pool[int(rng.integers(0, len(pool)))]
                if rng.random() < 0.5:
                    lit = lit[: int(rng.integers(0, len(lit) + 1))] + (b"" if rng.random() < 0.5 else b"m")
                return (["EQ", "NEQ", "LT", "GT", "LEQ", "GEQ"][int(rng.integers(0, 6))], c, lit)
            leaves.append(strcmp)

            def like_dev(c=c):
                pool = [row[c] for row in rows[:80] if row[c] is not None and len(row[c]) < 60] or [b"abc"]
                src = pool[int(rng.integers(0, len(pool)))]
                pat = bytearray()
                for ch in src[: int(rng.integers(1, 12))]:
                    x = rng.random()
                    if x < 0.2:
                        pat += b"%"
                    elif x < 0.3:
                        pat += b"_"
                    elif ch not in b"%_":
                        pat.append(ch)
                if rng.random() < 0.7:
                    pat += b"%"
                return (["LIKE", "NOT_LIKE"][int(rng.integers(0, 2))], c, bytes(pat))
            leaves.append(like_dev)

            # a predicate the caller evaluates itself (here: contains an 'a'): handed over as a bitmap
            def like(c=c):
                mask = np.array([row[c] is not None and b"a" in row[c] for row in rows], dtype=bool)
                return ("BITMAP", np.packbits(mask, bitorder="little") if n else np.zeros(0, np.uint8))
            leaves.append(like)

    def build(d):
        if d == 0 or rng.random() < 0.3:
            return [leaves[int(rng.integers(0, len(leaves)))]()]
        k = rng.random()
        if k < 0.2:
            return build(d - 1) + [("NOT",)]
        return build(d - 1) + build(d - 1) + [("AND",) if k < 0.6 else ("OR",)]

    return build(depth)
