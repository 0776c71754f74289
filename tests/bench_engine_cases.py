"""The reference's own benchmark list, as data.

Every case of `benchmarks/bench_engine.mojo:578-1100` (name, operation, pattern, input text), so
that the product can be checked and timed on exactly what the reference times.  The reference runs
one text per call on one CPU core; here a case's text becomes a batch (see `case_batch`).  The
`ct_*` cases (comptime API) share pattern and text with their `rtapi_*` twins and are listed as
aliases.  Operations: "match_first" = `CompiledRegex.match_first`, "search" = `match_next(text, 0)`,
"findall" = `match_all`, "is_match", "sub".
"""
from typing import Callable, List, NamedTuple, Optional


def _alphabet(n: int) -> bytes:                       # bench_engine.mojo:12-24
    a = b"abcdefghijklmnopqrstuvwxyz"
    return a * (n // 26) + a[: n % 26]


def _phones(n: int) -> bytes:                         # bench_engine.mojo:27-49
    forms = [b"555-123-4567", b"(555) 123-4567", b"555.123.4567", b"5551234567", b"+1-555-123-4567",
             b"1-555-123-4568", b"(555)123-4569", b"555 123 4570"]
    return b"".join(b" Contact us at " + forms[i % 8] + b" or email support@company.com for assistance. "
                    for i in range(n))


def _national(n: int) -> bytes:                       # bench_engine.mojo:52-75
    ids = [b"305200123456", b"505601234567", b"274212345678", b"305912345678", b"212345672890",
           b"312345672890", b"412345672890", b"512345672890", b"1234567890", b"30520"]
    return b"".join(b" ID: " + ids[i % 10] + b" Status: ACTIVE " for i in range(n))


TEXT_1000 = _alphabet(1000) + b"hello world"
TEXT_10000 = _alphabet(10000) + b"hello world"
TEXT_RANGE = _alphabet(10000) + b"0123456789"
TEXT_DIGITS = b"0123456789" * 1000 + b"abcdefghijklmnopqrstuvwxyz"
SHORT = b"hello world this is a test with hello again and hello there"
MEDIUM = SHORT * 100
LONG = SHORT * 1000
# adjacent string literals concatenate before `*` applies (as in Python)
EMAIL = (b"test@example.com user@test.org admin@example.com support@example.com"
         b" no-reply@example.com") * 50
TEXT_SHORT = b"say hello world, order ab12 shipped, ref 1234"
PHONE = _phones(1000)
SERIAL = (b"Serial: ABC1234-DEF5678-GHI9012 Model: XYZ123-ABC456-DEF789 "
          b"Part: MNO345-PQR678-STU901 Code: VWX234-YZA567-BCD890 ") * 50
DATETIME = (b"2024-01-15 14:30:25.123 2024-02-28 09:45:30.456 "
            b"2024-03-10 16:20:15.789 2024-04-05 11:35:40.012 ") * 100
STRUCTURED = (b"Record: USER12345-DEPT678-LOC901-ID234 Status:"
              b" ACTIVE567-FLAG890-CODE123 Transaction: TXN9876-AMT543-FEE210-TAX087"
              b" Reference: REF1357-NUM246-CHK802 ") * 75
OPTIMIZATION = (b"Transaction: TXN12345-DEPT678-LOC90123-ID4567 Status:"
                b" ACTIVE12-FLAG890-CODE1234 Reference: REF13579-NUM24680-CHK80246"
                b" Product: PROD123-CAT456-TYPE789-SUB012 ") * 100
TOLL_FREE = (b"Call 8001234567 or 9005551234 for assistance. Try 8775559999 or"
             b" 8006667777.") * 100
FILLER = b"The quick brown fox jumps over the lazy dog. " * 40
NATIONAL_PATTERN = (
    rb"(?:3052(?:0[0-8]|[1-9]\d)|5056(?:[0-35-9]\d|4[0-68]))\d{4}|(?:2742|305[3-9]|472[247-9]|505[2-57-9]|983[2-47-9])\d{6}|"
    rb"(?:2(?:0[1-35-9]|1[02-9]|2[03-57-9]|3[1459]|4[08]|5[1-46]|6[0279]|7[0269]|8[13])|3(?:0[1-47-9]|1[02-9]|2[0135-79]|3[0-24679]|"
    rb"4[167]|5[0-2]|6[01349]|8[056])|4(?:0[124-9]|1[02-579]|2[3-5]|3[0245]|4[023578]|58|6[349]|7[0589]|8[04])|5(?:0[1-47-9]|"
    rb"1[0235-8]|20|3[0149]|4[01]|5[179]|6[1-47]|7[0-5]|8[0256])|6(?:0[1-35-9]|1[024-9]|2[03689]|3[016]|4[0156]|5[01679]|6[0-279]|"
    rb"78|8[0-29])|7(?:0[1-46-8]|1[2-9]|2[04-8]|3[0-247]|4[037]|5[47]|6[02359]|7[0-59]|8[156])|8(?:0[1-68]|1[02-8]|2[0168]|"
    rb"3[0-2589]|4[03578]|5[046-9]|6[02-5]|7[028])|9(?:0[1346-9]|1[02-9]|2[0589]|3[0146-8]|4[01357-9]|5[12469]|7[0-389]|"
    rb"8[04-69]))[2-9]\d{6}")
# the third alternative of NATIONAL_PATTERN on its own (bench_engine.mojo:1105)
NANPA_PATTERN = NATIONAL_PATTERN[NATIONAL_PATTERN.index(b"|(?:2(?:0[1-35-9]") + 1:]
NANPA_TEXT = b"Call 6502530000 or 2125551234 or 9175559876. " * 50


class Case(NamedTuple):
    name: str
    op: str
    pattern: bytes
    text: bytes
    repl: Optional[bytes] = None
    count: int = 0
    aliases: tuple = ()


CASES: List[Case] = [
    Case("literal_match_short", "search", b"hello", TEXT_1000),
    Case("literal_match_long", "search", b"hello", TEXT_10000, aliases=("rtapi_literal_long", "ct_literal_long")),
    Case("rtapi_literal_short", "search", b"hello", TEXT_SHORT, aliases=("ct_literal_short",)),
    Case("rtapi_char_class_short", "search", b"[0-9]+", TEXT_SHORT, aliases=("ct_char_class_short",)),
    Case("rtapi_multi_class_short", "search", b"[a-z]+[0-9]+", TEXT_SHORT, aliases=("ct_multi_class_short",)),
    Case("wildcard_match_any", "match_first", b".*", TEXT_10000),
    Case("quantifier_zero_or_more", "match_first", b"a*", TEXT_10000),
    Case("quantifier_one_or_more", "match_first", b"a+", TEXT_10000),
    Case("quantifier_zero_or_one", "match_first", b"a?", TEXT_10000),
    Case("range_lowercase", "match_first", b"[a-z]+", TEXT_RANGE),
    Case("range_digits", "search", b"[0-9]+", TEXT_RANGE),
    Case("range_alphanumeric", "match_first", b"[a-zA-Z0-9]+", TEXT_RANGE),
    Case("predefined_digits", "search", rb"\d+", TEXT_RANGE),
    Case("predefined_word", "match_first", rb"\w+", TEXT_RANGE),
    Case("anchor_start", "match_first", b"^abc", TEXT_10000),
    Case("anchor_end", "match_first", b"xyz$", TEXT_10000),
    Case("alternation_simple", "match_first", b"a|b|c", TEXT_10000),
    Case("group_alternation", "match_first", b"(a|b)", TEXT_10000),
    Case("large_8_alternations", "search", b"(apple|banana|cherry|date|elderberry|fig|grape|honey)",
         b"I love eating apple and banana and cherry and date and elderberry and fig and grape with honey"),
    Case("deep_nested_groups_depth4", "search", b"(?:(?:(?:a|b)|(?:c|d))|(?:(?:e|f)|(?:g|h)))",
         b"Testing deep nested patterns with abcdefgh characters"),
    Case("literal_heavy_alternation", "search",
         b"(user123|admin456|guest789|root000|test111|demo222|sample333|client444)",
         b"Login attempts: user123 failed, admin456 success, guest789 failed, root000 success, test111 pending,"
         b" demo222 active, sample333 inactive, client444 locked"),
    Case("complex_group_5_children", "search", b"(hello|world|test|demo|sample)[0-9]{3}[a-z]{2}",
         b"Found: hello123ab, world456cd, test789ef, demo012gh, sample345ij in the data"),
    Case("match_all_simple", "findall", b"hello", MEDIUM),
    Case("match_all_digits", "findall", b"[0-9]+", TEXT_RANGE * 10),
    Case("literal_prefix_short", "findall", b"hello.*", SHORT),
    Case("literal_prefix_long", "findall", b"hello.*", LONG),
    Case("required_literal_short", "findall", rb".*@example\.com", EMAIL),
    Case("no_literal_baseline", "match_first", b"[a-z]+", MEDIUM),
    Case("alternation_common_prefix", "match_first", b"(hello|help|helicopter)", MEDIUM),
    Case("complex_email", "findall", rb"[a-zA-Z0-9._%+-]+@[a-zA-Z0-9.-]+\.[a-zA-Z]{2,}",
         b"Contact: john@example.com, support@test.org, admin@company.net" * 20),
    Case("complex_number", "findall", rb"[0-9]+\.[0-9]+", b"Price: $123.45, Quantity: 67, Total: $890.12, Tax: 15.5%" * 100),
    Case("simple_phone", "findall", rb"\d{3}-\d{3}-\d{4}", PHONE),
    Case("flexible_phone", "findall", rb"\(?\d{3}\)?[\s.-]?\d{3}[\s.-]?\d{4}", PHONE),
    Case("multi_format_phone", "findall", rb"\(?\d{3}\)?[\s.-]\d{3}[\s.-]\d{4}|\d{3}-\d{3}-\d{4}|\d{10}", PHONE),
    Case("phone_validation", "match_first", rb"^\+?1?[\s.-]?\(?([2-9]\d{2})\)?[\s.-]?([2-9]\d{2})[\s.-]?(\d{4})$",
         b"234-567-8901"),
    Case("dfa_simple_phone", "findall", b"[0-9]{3}-[0-9]{3}-[0-9]{4}", PHONE, aliases=("smart_phone_primary",)),
    Case("dfa_paren_phone", "findall", rb"\([0-9]{3}\) [0-9]{3}-[0-9]{4}", PHONE),
    Case("dfa_dot_phone", "findall", rb"[0-9]{3}\.[0-9]{3}\.[0-9]{4}", PHONE),
    Case("dfa_digits_only", "findall", b"[0-9]{10}", PHONE),
    Case("pure_dfa_dash", "findall", b"555-123-4567",
         b"Contact us at 555-123-4567 or call (555) 123-4567. Our fax is 555.123.4567."),
    Case("pure_dfa_paren", "findall", rb"\(555\) 123-4567",
         b"Contact us at 555-123-4567 or call (555) 123-4567. Our fax is 555.123.4567."),
    Case("pure_dfa_dot", "findall", rb"555\.123\.4567",
         b"Contact us at 555-123-4567 or call (555) 123-4567. Our fax is 555.123.4567."),
    Case("national_phone_validation", "findall", NATIONAL_PATTERN, _national(500)),
    Case("toll_free_simple", "findall", rb"[89]00\d{6}", TOLL_FREE),
    Case("toll_free_complex", "findall", rb"8(?:00|33|44|55|66|77|88)[2-9]\d{6}", TOLL_FREE),
    Case("single_quantifier_digits", "findall", b"[0-9]{4}", SERIAL),
    Case("single_quantifier_alpha", "findall", b"[A-Z]{3}", SERIAL),
    Case("dual_quantifiers", "findall", b"[A-Z]{3}[0-9]{4}", SERIAL),
    Case("triple_quantifiers", "findall", b"[A-Z]{3}[0-9]{4}-[A-Z]{3}[0-9]{3}", SERIAL),
    Case("quad_quantifiers", "findall", b"[A-Z]{3}[0-9]{4}-[A-Z]{3}[0-9]{3}-[A-Z]{3}[0-9]{3}", SERIAL),
    Case("range_quantifiers", "findall", b"[A-Z]{2,4}[0-9]{3,5}", SERIAL),
    Case("mixed_range_quantifiers", "findall", b"[A-Z]{1,3}-[0-9]{2,4}-[A-Z]{2,3}[0-9]{3,4}", SERIAL),
    Case("datetime_quantifiers", "findall", rb"[0-9]{4}-[0-9]{2}-[0-9]{2} [0-9]{2}:[0-9]{2}:[0-9]{2}\.[0-9]{3}", DATETIME),
    Case("flexible_datetime", "findall", b"[0-9]{4}-[0-9]{1,2}-[0-9]{1,2} [0-9]{1,2}:[0-9]{2}:[0-9]{2}", DATETIME),
    Case("dense_quantifiers", "findall", b"[A-Z]{2}[0-9]{5}-[A-Z]{4}[0-9]{3}-[A-Z]{3}[0-9]{3}-[A-Z]{2}[0-9]{3}", STRUCTURED),
    Case("ultra_dense_quantifiers", "findall",
         b"[A-Z]{1,2}[0-9]{3,5}-[A-Z]{2,4}[0-9]{2,4}-[A-Z]{1,3}[0-9]{2,4}-[A-Z]{2,3}[0-9]{2,3}", STRUCTURED),
    Case("grouped_quantifiers", "findall", b"([A-Z]{3}[0-9]{4})-([A-Z]{3}[0-9]{3})", SERIAL),
    Case("alternation_quantifiers", "findall", b"([A-Z]{2,3}[0-9]{3,4})|([0-9]{4}-[A-Z]{3})", STRUCTURED),
    Case("optimize_range_quantifier", "findall", b"a{2,4}", b"aaaabbbbccccdddd" * 500),
    Case("optimize_multiple_quantifiers", "findall", b"[A-Z]{3}[0-9]{4}-[A-Z]{3}[0-9]{3}-[A-Z]{2}[0-9]{2}", OPTIMIZATION),
    Case("optimize_phone_quantifiers", "findall", b"[0-9]{3}-[0-9]{3}-[0-9]{4}",
         b"Call 555-123-4567 or 800-555-1234 or 900-876-5432 for help. " * 200),
    Case("optimize_large_quantifiers", "findall", b"[A-Z]{10,20}[0-9]{15,25}",
         b"PREFIX" + b"A" * 15 + b"1" * 20 + b"SUFFIX " * 50),
    Case("optimize_extreme_quantifiers", "findall", b"a{1}b{2}c{3}d{4}e{5}f{6}g{7}h{8}",
         b"abcccddddeeeeeffffffggggggghhhhhhhhSEPARATOR" * 20),
    Case("is_match_lowercase", "is_match", b"[a-z]+", TEXT_RANGE),
    Case("is_match_digits", "is_match", b"[0-9]+", TEXT_DIGITS),
    Case("is_match_alphanumeric", "is_match", b"[a-zA-Z0-9]+", TEXT_RANGE),
    Case("is_match_predefined_digits", "is_match", rb"\d+", TEXT_DIGITS),
    Case("is_match_predefined_word", "is_match", rb"\w+", TEXT_RANGE),
    Case("sub_literal", "sub", b"hello", SHORT * 20, repl=b"REPLACED"),
    Case("sub_digits", "sub", rb"\d{3}-\d{3}-\d{4}", PHONE, repl=b"XXX-XXX-XXXX"),
    Case("sub_char_class", "sub", b"[0-9]+", PHONE, repl=b"#"),
    Case("sub_whitespace", "sub", rb"\s+", b"  hello   world   foo   bar   baz  " * 100, repl=b" "),
    # the reference's runner takes no count, so despite its name this replaces every match
    Case("sub_limited_count", "sub", b"hello", SHORT * 100, repl=b"HI"),
    Case("sub_group_phone_fmt", "sub", rb"(\d{3})(\d{3})(\d{4})", b"Call 6502530000 or 4155551234 today. " * 100,
         repl=rb"\1-\2-\3"),
    Case("sub_group_date_fmt", "sub", rb"(\d{4})-(\d{2})-(\d{2})",
         b"Event on 2026-04-12 and 2025-12-25 and 2024-01-01. " * 50, repl=rb"\2/\3/\1"),
    Case("sub_group_word_swap", "sub", rb"(\w+) (\w+)", b"hello world foo bar baz qux " * 50, repl=rb"\2 \1"),
    Case("sparse_phone_findall", "findall", rb"\d{3}-\d{3}-\d{4}", (FILLER + b"Call 555-123-4567 now. ") * 20),
    Case("sparse_phone_search", "search", rb"\(\d{3}\)\s\d{3}-\d{4}", FILLER * 50 + b"(555) 123-4567" + FILLER * 50),
    Case("sparse_email_findall", "findall", rb"[a-zA-Z0-9._%+-]+@[a-zA-Z0-9.-]+\.[a-zA-Z]{2,}",
         (FILLER + b"Contact admin@example.com for details. ") * 10),
    Case("sparse_flex_phone_findall", "findall", rb"\(?\d{3}\)?[\s.-]?\d{3}[\s.-]?\d{4}",
         (FILLER + b"Reach us at (555) 123-4567 today. ") * 10),
    Case("nanpa_findall", "findall", NANPA_PATTERN, NANPA_TEXT),
    Case("nanpa_search", "search", NANPA_PATTERN, NANPA_TEXT),
    Case("nanpa_match_first", "match_first", NANPA_PATTERN, b"6502530000"),
]


def case_rows(case: Case, rows: int) -> List[bytes]:
    """`rows` texts derived from a case's text: the text itself, then rotations of it by multiples of
    37 bytes (same bytes, every alignment, matches cut at the seam), so that the 64 lanes of a
    wavefront do not walk in lockstep."""
    t = case.text
    out = [t]
    for i in range(1, rows):
        k = (37 * i) % max(len(t), 1)
        out.append(t[k:] + t[:k])
    return out

