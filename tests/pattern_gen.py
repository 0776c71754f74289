"""Random pattern generator shared by the CPU table-equality fuzz and the GPU result fuzz.
Grammar-directed so that most patterns parse and hit the reference's shape recognisers: literals,
classes, predefined classes, quantifiers, groups, alternations, anchors, plus some junk."""
import random

ATOMS = ["a", "b", "c", "x", "y", "z", "0", "1", "9", " ", "-", "@", "\\.", "\\d", "\\w", "\\s", ".",
         "[a-z]", "[0-9]", "[a-c]", "[A-Z]", "[abc]", "[^0-9]", "[^a-z]", "[a-z0-9]", "[a-zA-Z]", "[xyz]",
         "[0-9a-f]", "[ab]", "[.-]", "[\\s.-]", "[\\d]", "[a\\-z]", "[+]"]
QUANTS = ["", "", "", "", "+", "*", "?", "{2}", "{3}", "{1,3}", "{2,}", "{0,2}", "{1,}", "{4}"]
WORDS = ["foo", "bar", "baz", "hello", "cat", "dog", "ab", "abc", "xy", "http", "id", "no"]


def _atom(r):
    return r.choice(ATOMS)


def _seq(r, depth):
    n = r.choice([1, 1, 2, 2, 3, 4])
    out = []
    for _ in range(n):
        k = r.random()
        if k < 0.62 or depth >= 2:
            out.append(_atom(r) + r.choice(QUANTS))
        elif k < 0.75:
            out.append(r.choice(WORDS))
        elif k < 0.9:
            out.append("(" + _alt(r, depth + 1) + ")" + r.choice(QUANTS))
        else:
            out.append("(?:" + _alt(r, depth + 1) + ")" + r.choice(QUANTS))
    return "".join(out)


def _alt(r, depth):
    k = r.choice([1, 1, 1, 2, 2, 3, 4])
    parts = []
    for _ in range(k):
        parts.append(r.choice(WORDS) if r.random() < 0.4 else _seq(r, depth))
    return "|".join(parts)


def gen_pattern(r: random.Random) -> str:
    p = _alt(r, 0)
    k = r.random()
    if k < 0.08:
        p = "^" + p
    elif k < 0.14:
        p = p + "$"
    elif k < 0.2:
        p = "^" + p + "$"
    elif k < 0.23:   # junk: unbalanced / odd tokens the parser must reject (or accept) identically
        i = r.randrange(len(p) + 1)
        p = p[:i] + r.choice(["(", ")", "[", "]", "{", "}", "|", "\\", "*", "+", "?", "^", "$"]) + p[i:]
    return p


def patterns(seed: int, n: int):
    r = random.Random(seed)
    seen, out = set(), []
    while len(out) < n:
        p = gen_pattern(r)
        if p not in seen and len(p) <= 60:
            seen.add(p)
            out.append(p)
    return out


# ---- second generator: the shapes the first one rarely reaches ---------------------------------------------------
# literal-led and literal-tailed '.*' patterns (NFAEngine's prefilter and fast paths), alternations of words,
# escaped specials, the upper-case predefined classes, larger counted repetitions, phone / date / e-mail style
# concatenations, anchors inside alternations.
ATOMS2 = ["\\d", "\\w", "\\s", "\\D", "\\W", "\\S", "[a-z]", "[A-Z]", "[0-9]", "[a-zA-Z0-9._%+-]", "[\\s.-]", "[^ ]", "[^@]",
          "\\.", "\\(", "\\)", "\\[", "\\+", "\\\\", "\\t", "\\n", "-", "@", ":", "/", " ", "a", "e", "o", "x", "0", "."]
QUANTS2 = ["", "", "", "+", "*", "?", "{2}", "{3}", "{4}", "{2,4}", "{3,}", "{1,2}", "{5,10}", "{10}", "{0,1}"]
WORDS2 = ["hello", "world", "foo", "bar", "foobar", "http", "https", "com", "org", "example", "user", "error", "id", "GET",
          "a", "ab", "abc", "the", "cat", "42"]


def _piece2(r):
    k = r.random()
    if k < 0.55:
        return r.choice(ATOMS2) + r.choice(QUANTS2)
    if k < 0.8:
        return r.choice(WORDS2)
    if k < 0.9:
        inner = "|".join(r.choice(WORDS2) if r.random() < 0.6 else _piece2(r) + _piece2(r) for _ in range(r.choice([2, 2, 3, 4])))
        return r.choice(["(", "(?:"]) + inner + ")" + r.choice(QUANTS2)
    return "(" + "".join(_piece2(r) for _ in range(r.choice([1, 2, 3]))) + ")" + r.choice(["", "", "?", "+", "*", "{2}"])


def gen_pattern2(r: random.Random) -> str:
    style = r.random()
    body = "".join(_piece2(r) for _ in range(r.choice([1, 2, 2, 3, 3, 4, 5])))
    if style < 0.15:
        p = r.choice(WORDS2) + r.choice([".*", ".+", ".*", "\\s*"]) + body
    elif style < 0.3:
        p = body + r.choice([".*", ".+"]) + r.choice(WORDS2)
    elif style < 0.4:
        p = ".*" + body
    elif style < 0.5:
        p = "|".join(r.choice(WORDS2) for _ in range(r.choice([2, 3, 5, 8])))
    elif style < 0.6:
        p = body + "|" + "".join(_piece2(r) for _ in range(r.choice([1, 2])))
    else:
        p = body
    k = r.random()
    if k < 0.1:
        p = "^" + p
    elif k < 0.2:
        p = p + "$"
    elif k < 0.27:
        p = "^" + p + "$"
    elif k < 0.3:
        p = p.replace("|", "$|^", 1) if "|" in p else "^" + p
    return p


def patterns2(seed: int, n: int):
    r = random.Random(seed)
    seen, out = set(), []
    while len(out) < n:
        p = gen_pattern2(r)
        if p not in seen and len(p) <= 70:
            seen.add(p)
            out.append(p)
    return out
