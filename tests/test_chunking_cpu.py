"""The reference's own chunker unit tests (src/embeddings/chunking.rs:420-671) against the Python mirror that
`RetrievalEngine.index_memory` uses."""
from shodh_memory_amd.chunking import MODEL_TOKEN_WINDOW, SPECIAL_TOKEN_OVERHEAD, ChunkConfig, chunk_text


def word_counter(text):
    return len(text.split()) + SPECIAL_TOKEN_OVERHEAD


def cfg(max_tokens, overlap):
    return ChunkConfig(max_tokens=max_tokens, overlap_tokens=overlap)


def test_chunk_budget_never_exceeds_model_window():
    assert ChunkConfig().max_tokens <= MODEL_TOKEN_WINDOW and ChunkConfig.for_budget(5).max_tokens == 32


def test_short_text_single_chunk():
    r = chunk_text("This is a short text.", ChunkConfig(), word_counter)
    assert len(r.chunks) == 1 and not r.was_chunked and r.chunks[0] == "This is a short text."


def test_overlap_never_overflows_and_never_forces_a_word_split():
    config = cfg(22, 8)
    filler = " ".join("f%d" % i for i in range(1, 13))
    small = "Alpha beta gamma delta."
    big = " ".join("w%d" % i for i in range(1, 21))
    r = chunk_text("%s. %s %s." % (filler, small, big), config, word_counter)
    assert all(word_counter(c) <= config.max_tokens for c in r.chunks)
    assert any("w1 " in c and "w20" in c for c in r.chunks), r.chunks


def _forty():
    return " ".join("Sentence number %d contains unique information." % i for i in range(1, 41))


def test_every_chunk_fits_token_budget_and_no_content_lost():
    config = cfg(30, 8)
    r = chunk_text(_forty(), config, word_counter)
    assert r.was_chunked and all(word_counter(c) <= config.max_tokens for c in r.chunks)
    for i in range(1, 41):
        assert any(("number %d " % i) in c or ("number %d contains" % i) in c for c in r.chunks), i


def test_unique_markers_beginning_middle_end_searchable():
    text = "%s %s %s %s %s" % ("ALPHA_BEGINNING_MARKER is a unique identifier at the start.", "This is filler content to push things apart. " * 20,
                              "BETA_MIDDLE_MARKER represents content in the center of the document.", "More filler content for separation between sections. " * 20,
                              "GAMMA_END_MARKER signifies the conclusion of this memory content.")
    r = chunk_text(text, cfg(40, 8), word_counter)
    assert r.was_chunked and all(any(m in c for c in r.chunks) for m in ("ALPHA_BEGINNING", "BETA_MIDDLE", "GAMMA_END"))


def test_sentence_boundaries_respected():
    text = "First sentence here. Second sentence follows on. Third sentence ends it. Fourth sentence too. Fifth sentence closes."
    r = chunk_text(text, cfg(12, 2), word_counter)
    assert len(r.chunks) > 1 and all(c.rstrip()[-1] in ".!?" for c in r.chunks[:-1]), r.chunks


def test_overlap_carries_small_trailing_sentence():
    text = " ".join("Overlap test sentence %d has exactly eight words." % i for i in range(1, 7))
    r = chunk_text(text, cfg(20, 10), word_counter)
    assert len(r.chunks) >= 2
    for prev, nxt in zip(r.chunks, r.chunks[1:]):
        last_sentence = prev.split(". ")[-1].strip()
        assert last_sentence.rstrip(".") in nxt, (prev, nxt)


def test_dialogue_turns_not_split_when_they_fit():
    text = ("Alice: I went to the market this morning and bought fresh vegetables.\nBob: That sounds great, did you find good tomatoes there?\n"
            "Alice: Yes, and I also picked up some basil for the sauce.\nBob: Perfect, let us cook dinner together tonight then.")
    r = chunk_text(text, cfg(25, 4), word_counter)
    for c in r.chunks:
        for name in ("Alice:", "Bob:"):
            i = c.find(name)
            while i != -1:
                assert i == 0 or c[i - 1] == "\n", c
                i = c.find(name, i + 1)


def test_oversized_single_sentence_is_word_split():
    r = chunk_text(("word " * 100).strip(), cfg(12, 2), word_counter)
    assert len(r.chunks) >= 10 and all(word_counter(c) <= 12 for c in r.chunks) and sum(len(c.split()) for c in r.chunks) == 100


def test_pathological_single_token_run_is_bisected():
    char_counter = lambda t: (len(t) + 3) // 4 + SPECIAL_TOKEN_OVERHEAD
    r = chunk_text("x" * 400, cfg(10, 2), char_counter)
    assert all(char_counter(c) <= 10 for c in r.chunks) and sum(len(c) for c in r.chunks) == 400
    r = chunk_text("é" * 300, cfg(10, 2), char_counter)                       # bisection lands on char boundaries
    assert "".join(r.chunks) == "é" * 300 and all(char_counter(c) <= 10 for c in r.chunks)


def test_trailing_fragment_never_lost_and_never_overflows():
    text = ("This first sentence is exactly nine words long okay.                     This second sentence is also exactly nine words long. Tiny tail here")
    r = chunk_text(text, cfg(20, 4), word_counter)
    assert any("Tiny tail here" in c for c in r.chunks) and all(word_counter(c) <= 20 for c in r.chunks)


def test_empty_text_single_empty_chunk():
    r = chunk_text("", ChunkConfig(), word_counter)
    assert len(r.chunks) == 1 and not r.was_chunked
