"""mojo_regex_amd -- MI355X-native batch regex matcher.

Python host layer above the C ABI (include/mrx.h, libmrx_hip.so).  It mirrors
the reference's public interface for the matching hot path -- same names and
argument meaning, a *batch* of texts instead of one text:

    reference (src/regex/matcher.mojo)            here
    ------------------------------------------    ------------------------------
    compile_regex(pattern)            :1292       compile_regex(pattern)
    CompiledRegex.match_first(text)   :1049       CompiledRegex.match_first(texts)
    CompiledRegex.match_next(text)    :1064       CompiledRegex.match_next(texts)
    CompiledRegex.match_all(text)     :1078       CompiledRegex.match_all(texts)
    CompiledRegex.is_match(text)      :1104       CompiledRegex.is_match(texts)
    CompiledRegex.sub(repl, text)     :1118       CompiledRegex.sub(repl, texts)
    CompiledRegex.get_stats()         :1139       CompiledRegex.get_stats()
    match_first / search / findall    :1325-1415  match_first / search / findall
    split / sub                       :1357,1857  split / sub
    clear_regex_cache()               :1318       clear_regex_cache()

All matching runs in the HIP kernels of libmrx_hip.so.  There is no CPU
fallback: if the library is missing or no GPU is usable, calls raise.
(The directory is named mojo_regex_amd because Python cannot import a package
whose name contains '-'.)
"""
from .api import (  # noqa: F401
    CompiledRegex,
    DeviceBatch,
    MrxError,
    RegexSyntaxError,
    UnsupportedPattern,
    clear_regex_cache,
    compile_regex,
    findall,
    library_path,
    load_library,
    match_first,
    pack_texts,
    search,
    split,
    sub,
)
