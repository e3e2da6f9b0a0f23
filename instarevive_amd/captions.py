"""Caption cleaning of the prompt producer: T5Embedder.text_preprocessing / clean_caption (diffusion/model/t5.py:106-233).

The reference cleans every caption TWICE with a fixed sequence of rewrites (URL / HTML / CJK / punctuation / id / boilerplate removal)
before tokenising it; the tokens, hence the prompt embedding the restoration path consumes, depend on the exact sequence. It is restated
here as data: a table of (pattern, replacement) steps applied in order, with the two non-regex steps as functions:

  * HTML -> text (reference: BeautifulSoup(caption, features='html.parser').text, t5.py:134): `strip_tags`, the standard library's
    html.parser collecting the text nodes - the same parser family bs4 drives with features='html.parser';
  * basic_clean (t5.py:118-121: ftfy.fix_text + two html.unescape): `basic_clean` below applies the deterministic parts of ftfy's default
    fix_text (HTML entities, Latin ligatures, full-width forms, curly quotes, line breaks, control characters, NFC). ftfy's mojibake
    repair is restated for its dominant case only (`fix_mojibake`: UTF-8 read as cp1252 / Latin-1, up to three layers; **parity unpinned** -
    tools/repin_with_diffusers.py compares it with ftfy where ftfy exists). Neither ftfy nor bs4 exists in this image, and the fixture
    tests/golden/captions.json was generated from the reference's own clean_caption with pass-through stand-ins for exactly those two
    calls (recorded in the fixture), on captions they leave unchanged.
"""
import html
import re
import unicodedata
import urllib.parse as ul
from html.parser import HTMLParser

# t5.py:15-16 (class attribute bad_punct_regex)
BAD_PUNCT = re.compile(r'[' + '#®•©™&@·º½¾¿¡§~' + r'\)' + r'\(' + r'\]' + r'\[' + r'\}' + r'\{' + r'\|' + '\\' + r'\/' + r'\*' + r']{1,}')

_URL_TLDS = r"(?:com|co|ru|net|org|edu|gov|it)"
# steps before the HTML stripping (t5.py:124-132)
_PRE = [
    (r"<person>", "person"),
    (r"\b((?:https?:(?:\/{1,3}|[a-zA-Z0-9%])|[a-zA-Z0-9.\-]+[.]" + _URL_TLDS + r"[\w/-]*\b\/?(?!@)))", ""),
    (r"\b((?:www:(?:\/{1,3}|[a-zA-Z0-9%])|[a-zA-Z0-9.\-]+[.]" + _URL_TLDS + r"[\w/-]*\b\/?(?!@)))", ""),
]
# t5.py:137-197: nicknames, CJK blocks, dashes, quotes, entities, ids, file names, repeated punctuation
_MID = [
    (r"@[\w\d]+\b", ""),
    (r"[\u31c0-\u31ef]+", ""), (r"[\u31f0-\u31ff]+", ""), (r"[\u3200-\u32ff]+", ""), (r"[\u3300-\u33ff]+", ""),
    (r"[\u3400-\u4dbf]+", ""), (r"[\u4dc0-\u4dff]+", ""), (r"[\u4e00-\u9fff]+", ""),
    (r"[\u002D\u058A\u05BE\u1400\u1806\u2010-\u2015\u2E17\u2E1A\u2E3A\u2E3B\u2E40\u301C\u3030\u30A0\uFE31\uFE32\uFE58\uFE63\uFF0D]+", "-"),
    (r"[`´«»“”¨]", '"'), (r"[‘’]", "'"),
    (r"&quot;?", ""), (r"&amp", ""),
    (r"\d{1,3}\.\d{1,3}\.\d{1,3}\.\d{1,3}", " "),
    (r"\d:\d\d\s+$", ""),
    (r"\\n", " "),
    (r"#\d{1,3}\b", ""), (r"#\d{5,}\b", ""), (r"\b\d{6,}\b", ""),
    (r"[\S]+\.(?:png|jpg|jpeg|bmp|webp|eps|pdf|apk|mp4)", ""),
    (r"[\"\']{2,}", '"'), (r"[\.]{2,}", " "),
    (BAD_PUNCT, " "),
    (r"\s+\.\s+", " "),
]
# t5.py:206-224: after basic_clean
_POST = [
    (r"\b[a-zA-Z]{1,3}\d{3,15}\b", ""), (r"\b[a-zA-Z]+\d+[a-zA-Z]+\b", ""), (r"\b\d+[a-zA-Z]+\d+\b", ""),
    (r"(worldwide\s+)?(free\s+)?shipping", ""), (r"(free\s)?download(\sfree)?", ""), (r"\bclick\b\s(?:for|on)\s\w+", ""),
    (r"\b(?:png|jpg|jpeg|bmp|webp|eps|pdf|apk|mp4)(\simage[s]?)?", ""), (r"\bpage\s+\d+\b", ""),
    (r"\b\d*[a-zA-Z]+\d+[a-zA-Z]+\d+[a-zA-Z\d]*\b", " "),
    (r"\b\d+\.?\d*[xх×]\d+\.?\d*\b", ""),
    (r"\b\s+\:\s+", ": "), (r"(\D[,\./])\b", r"\1 "), (r"\s+", " "),
    # (t5.py:226 `caption.strip()` discards its result: no step)
    (r"^[\"\']([\w\W]+)[\"\']$", r"\1"), (r"^[\'\_,\-\:;]", ""), (r"[\'\_,\-\:\-\+]$", ""), (r"^\.\S+$", ""),
]
_PRE, _MID, _POST = ([(re.compile(p) if isinstance(p, str) else p, r) for p, r in t] for t in (_PRE, _MID, _POST))
_DASHES = re.compile(r"(?:\-|\_)")


class _Text(HTMLParser):
    def __init__(self):
        super().__init__(convert_charrefs=True)
        self.parts = []

    def handle_data(self, data):
        self.parts.append(data)


def strip_tags(text: str) -> str:
    """The text nodes of `text` parsed as HTML (character references resolved), concatenated."""
    p = _Text()
    p.feed(text)
    p.close()
    return "".join(p.parts)


_LIGATURES = {"\ufb00": "ff", "\ufb01": "fi", "\ufb02": "fl", "\ufb03": "ffi", "\ufb04": "ffl", "\ufb05": "ſt", "\ufb06": "st", "\u0132": "IJ", "\u0133": "ij"}
_QUOTES = {"\u2018": "'", "\u2019": "'", "\u201a": "'", "\u201b": "'", "\u201c": '"', "\u201d": '"', "\u201e": '"', "\u201f": '"'}
_CONTROL = re.compile("[\x00-\x08\x0b\x0e-\x1f\x7f\u200b-\u200f\u202a-\u202e\ufeff\ufff9-\ufffb]")


# UTF-8 text that was decoded as Windows-1252 / Latin-1 (once or twice): "cafÃ©", "itâ€™s", "Ã¢â‚¬â„¢". ftfy finds such spans with a badness
# heuristic over several single-byte encodings; restated here is the dominant case only: a run of two or more non-ASCII characters that
# (a) starts like the lead byte of a UTF-8 sequence seen through cp1252 (Ã Â â ...), (b) encodes to cp1252 / latin-1 bytes, and (c) those bytes
# ARE valid UTF-8. Text that is not mojibake fails (b) or (c) and is left alone ("éà" -> E9 E0 is not UTF-8).
_MOJIBAKE_RUN = re.compile(r"[\u00c2-\u00f4][^\x00-\x7f]+")
_C1_GAPS = {0x81, 0x8d, 0x8f, 0x90, 0x9d}   # bytes cp1252 leaves undefined: mis-decoders pass them through as U+0081 ... ("sloppy" cp1252)


def _redecode(run: str):
    out = bytearray()
    for ch in run:
        o = ord(ch)
        if o in _C1_GAPS or 0xa0 <= o <= 0xff:
            out.append(o)
        else:
            try:
                out += ch.encode("cp1252")
            except UnicodeEncodeError:
                return None
    try:
        return out.decode("utf-8")
    except UnicodeDecodeError:
        return None


def fix_mojibake(text: str) -> str:
    for _ in range(3):   # text mis-decoded more than once unwinds one layer per pass
        fixed = _MOJIBAKE_RUN.sub(lambda m: _redecode(m.group(0)) or m.group(0), text)
        if fixed == text:
            break
        text = fixed
    return text


def fix_text(text: str) -> str:
    """The steps of ftfy.fix_text's default configuration: the deterministic ones as ftfy defines them, its encoding repair in the restricted
    form of fix_mojibake above (see the module docstring)."""
    text = fix_mojibake(text)
    text = html.unescape(text)
    text = "".join(_LIGATURES.get(c, c) for c in text)
    text = "".join(unicodedata.normalize("NFKC", c) if "\uff01" <= c <= "\uff5e" or c == "\u3000" else c for c in text)   # full-width ASCII forms
    text = "".join(_QUOTES.get(c, c) for c in text)
    text = text.replace("\r\n", "\n").replace("\r", "\n").replace("\u2028", "\n").replace("\u2029", "\n").replace("\u0085", "\n")
    text = _CONTROL.sub("", text)
    return unicodedata.normalize("NFC", text)


def basic_clean(text: str, fix=fix_text) -> str:   # t5.py:118-121
    return html.unescape(html.unescape(fix(text))).strip()


def clean_caption(caption, html_to_text=strip_tags, fix=fix_text) -> str:
    """One pass of T5Embedder.clean_caption (t5.py:123-233)."""
    c = ul.unquote_plus(str(caption)).strip().lower()
    for rx, rep in _PRE:
        c = rx.sub(rep, c)
    c = html_to_text(c)
    for rx, rep in _MID:
        c = rx.sub(rep, c)
    if len(_DASHES.findall(c)) > 3:   # this-is-my-cute-cat / this_is_my_cute_cat
        c = _DASHES.sub(" ", c)
    c = basic_clean(c, fix)
    for rx, rep in _POST:
        c = rx.sub(rep, c)
    return c.strip()


def text_preprocessing(text: str, use_text_preprocessing: bool = True) -> str:
    """T5Embedder.text_preprocessing (t5.py:106-116): the cleaning runs twice, "the exact text cleaning as was in the training stage"."""
    if use_text_preprocessing:
        return clean_caption(clean_caption(text))
    return text.lower().strip()
