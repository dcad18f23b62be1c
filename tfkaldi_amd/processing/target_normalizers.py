"""Target normalisers: transcription -> space-separated symbols of a coder's alphabet (interface of the
reference's processing/target_normalizers.py; database and task dependent)."""

# spoken punctuation of the Aurora-4 / WSJ transcriptions: the leading sign is dropped, <NOISE> disappears
_AURORA4_WORDS = {
    ",COMMA": "COMMA", "\"DOUBLE-QUOTE": "DOUBLE-QUOTE", "!EXCLAMATION-POINT": "EXCLAMATION-POINT",
    "&AMPERSAND": "AMPERSAND", "'SINGLE-QUOTE": "SINGLE-QUOTE", "(LEFT-PAREN": "LEFT-PAREN",
    ")RIGHT-PAREN": "RIGHT-PAREN", "-DASH": "DASH", "-HYPHEN": "HYPHEN", "...ELLIPSIS": "ELLIPSIS",
    ".PERIOD": "PERIOD", "/SLASH": "SLASH", ":COLON": "COLON", ";SEMI-COLON": "SEMI-COLON", "<NOISE>": "",
    "?QUESTION-MARK": "QUESTION-MARK", "{LEFT-BRACE": "LEFT-BRACE", "}RIGHT-BRACE": "RIGHT-BRACE",
}


def aurora4_normalizer(transcription, alphabet):
    """Aurora-4 training transcriptions -> "<sos> c h a r s <space> ... <eos>" (reference
    target_normalizers.py:5-58): spoken-punctuation words lose their sign, everything is lower-cased and split
    into characters, blanks become <space>, characters outside `alphabet` become <unk>."""
    known = set(alphabet)
    words = (_AURORA4_WORDS.get(word, word) for word in transcription.split(" "))
    symbols = ["<sos>"]
    for ch in " ".join(words).lower():
        symbol = "<space>" if ch == " " else ch
        symbols.append(symbol if symbol in known else "<unk>")
    symbols.append("<eos>")
    return " ".join(symbols)
