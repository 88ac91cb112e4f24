"""Structural text chunker of `RetrievalEngine::index_memory` (src/embeddings/chunking.rs): dialogue turns > paragraphs >
sentences packed into chunks that each fit the embedder's token window, with one small trailing sentence carried over soft
boundaries. Pure host text logic; mirrored here so that `index_memory(memory_id, content=...)` splits long content exactly as
the reference does before the chunks reach the device encoder.

`counter(text)` is the TokenCounter contract (chunking.rs:65-69): the FULL encoded sequence length, special tokens included,
no truncation -- `Embedder.count_tokens`.
"""
import re
from collections import deque
from dataclasses import dataclass

MODEL_TOKEN_WINDOW = 128          # chunking.rs:60
SPECIAL_TOKEN_OVERHEAD = 2        # chunking.rs:63

_DIALOGUE_TURN = re.compile(r"^([A-Z][a-zA-Z0-9_\- ]{0,30})\s*:", re.M)       # chunking.rs:72-73
_SENTENCE_END = re.compile(r"""[.!?]+["')\]]*(?:\s+|\Z)""")                     # :77-78 (`$` without (?m) = end of text)
_PARAGRAPH = re.compile(r"\n\s*\n")                                             # :214-215
_HARD, _SOFT = 0, 1


@dataclass
class ChunkConfig:
    max_tokens: int = MODEL_TOKEN_WINDOW      # full-sequence tokens per chunk
    overlap_tokens: int = 24                  # content tokens a carried trailing sentence may have

    @classmethod
    def for_budget(cls, budget_tokens):
        return cls(max_tokens=max(budget_tokens, 32))


@dataclass
class ChunkResult:
    chunks: list
    original_length: int                      # bytes of the trimmed text (`text.len()`)
    was_chunked: bool


def is_dialogue_format(text):
    return _DIALOGUE_TURN.search(text) is not None


def _sat_sub(a, b):
    return a - b if a > b else 0


def _hard_split(text, content_budget, counter):                                 # chunking.rs:350-416
    budget = max(content_budget, 1)
    out, current, current_tokens = [], "", 0
    queue = deque(text.split())
    while queue:
        word = queue.popleft()
        word_tokens = _sat_sub(counter(word), SPECIAL_TOKEN_OVERHEAD)
        if word_tokens > budget:
            raw = word.encode("utf-8")
            split_at = len(raw) // 2
            while split_at > 0 and (raw[split_at] & 0xC0) == 0x80:              # is_char_boundary
                split_at -= 1
            if split_at == 0 or split_at == len(raw):
                if current:
                    out.append(current)
                    current, current_tokens = "", 0
                out.append(word)
                continue
            queue.appendleft(raw[split_at:].decode("utf-8"))
            queue.appendleft(raw[:split_at].decode("utf-8"))
            continue
        if current and current_tokens + word_tokens > budget:
            out.append(current)
            current, current_tokens = "", 0
        if current:
            current += " "
        current += word
        current_tokens += word_tokens
    if current:
        out.append(current)
    if not out:
        out.append(text)
    return out


def _split_sentences(span):                                                     # chunking.rs:264-281
    out = []
    for line in span.split("\n"):
        line = line.strip()
        if not line:
            continue
        last = 0
        for m in _SENTENCE_END.finditer(line):
            out.append(line[last:m.end()])
            last = m.end()
        if last < len(line):
            out.append(line[last:])
    return out


def _split_units(text, content_budget, counter):                                # chunking.rs:194-260
    if is_dialogue_format(text):
        starts = [m.start() for m in _DIALOGUE_TURN.finditer(text)]
        spans = []
        if starts and starts[0] > 0:
            spans.append(text[:starts[0]])
        for i, st in enumerate(starts):
            spans.append(text[st:starts[i + 1] if i + 1 < len(starts) else len(text)])
        if not spans:
            spans = [text]
    else:
        spans, last = [], 0
        for m in _PARAGRAPH.finditer(text):
            if m.start() > last:
                spans.append(text[last:m.start()])
            last = m.end()
        if last < len(text):
            spans.append(text[last:])
        if not spans:
            spans = [text]
    units = []                                                                   # (text, boundary, content tokens)
    for span in spans:
        span = span.strip()
        if not span:
            continue
        first_in_span = True
        for sentence in _split_sentences(span):
            sentence = sentence.strip()
            if not sentence:
                continue
            boundary = _HARD if first_in_span else _SOFT
            first_in_span = False
            tokens = _sat_sub(counter(sentence), SPECIAL_TOKEN_OVERHEAD)
            if tokens > content_budget:
                for i, piece in enumerate(_hard_split(sentence, content_budget, counter)):
                    units.append((piece, boundary if i == 0 else _SOFT, _sat_sub(counter(piece), SPECIAL_TOKEN_OVERHEAD)))
            else:
                units.append((sentence, boundary, tokens))
    return units


def _pack_units(units, config, content_budget):                                 # chunking.rs:285-345
    chunks, current, current_tokens, last_unit = [], "", 0, None
    for text, boundary, tokens in units:
        fits = current == "" or current_tokens + tokens <= content_budget
        if not fits:
            overlap = None
            if boundary == _SOFT and last_unit is not None:
                t = last_unit[1]
                if t <= config.overlap_tokens and t + tokens <= content_budget:
                    overlap = last_unit
                last_unit = None                                                 # `.take()` empties it either way
            chunks.append(current)
            current, current_tokens = "", 0
            if overlap is not None:
                current, current_tokens = overlap
        if current:
            current += "\n" if boundary == _HARD else " "
        current += text
        current_tokens += tokens
        last_unit = (text, tokens)
    if current:
        chunks.append(current)
    return chunks


def chunk_text(text, config, counter):
    """chunking.rs:148-184. Every returned chunk satisfies counter(chunk) <= config.max_tokens."""
    text = text.strip()
    original_length = len(text.encode("utf-8"))
    if counter(text) <= config.max_tokens:
        return ChunkResult([text], original_length, False)
    content_budget = _sat_sub(config.max_tokens, SPECIAL_TOKEN_OVERHEAD)
    chunks = _pack_units(_split_units(text, content_budget, counter), config, content_budget)
    verified = []
    for chunk in chunks:
        if counter(chunk) > config.max_tokens:
            verified.extend(_hard_split(chunk, content_budget, counter))
        else:
            verified.append(chunk)
    return ChunkResult(verified, original_length, len(verified) > 1)
