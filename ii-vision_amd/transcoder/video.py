"""Encode a sequence of images as an optimized stream of screen changes -- on the GPU.

Host-side mirror of the reference's transcoder/video.py: `Video` keeps the same
constructor, attributes (memory_map, aux_memory_map, pixelmap, update_priority,
aux_update_priority, out_of_work, frame_number) and methods (tick, encode_frame),
and `encode_frame` is still a lazy, never-ending generator of
(page, content, offsets) tuples.  The work -- diff weights, priority ranking, the
greedy selection loop and both MT19937 nonce streams -- runs in the gfx950 kernels
of csrc/iiv_prologue.hip / iiv_greedy.hip / iiv_team.hip through iiv_encode() (csrc/iiv_encode.hip, include/iivision.h).

State model: the device holds the live state while a generator runs; the host numpy
arrays (memory maps, update priorities, pixelmap.packed, out_of_work) and Python's /
NumPy's global RNG states are brought up to date *when somebody looks*: reading any of
the state attributes brings everything; starting another generator brings the RNG positions
and out_of_work (5 KB), and carries to the device whatever the caller changed in between --
arrays it was handed, a reseeded or advanced `random` / `np.random` -- so code that pokes
`video.memory_map.page_offset` or reseeds `random` between frames keeps working: the same
points at which the reference's caller (movie.py) looks.
`Video.STRICT_SYNC = True` restores the literal behaviour (every next() round-trips the
whole state, global RNG states included), at about 1.5 k opcodes per second.
Limitation of the default (lightweight) sync: WHILE a generator is live, the host's global
`random` / `np.random` states are the stale pre-launch ones.  A caller that draws from or reseeds
them between two next() calls of the same live generator (the reference's movie.py never does) and
then starts another generator has its draws applied on top of that stale state, which rewinds the
device's stream: the opcode stream then differs from the reference's.  Drawing or reseeding BETWEEN
generators -- after the previous one was abandoned, exhausted or its state read -- is supported;
for anything else set STRICT_SYNC.

Budget: a generator may be abandoned after any next() (movie.py:94-109 does so at
every frame and bank flip), and its side effects must then be exactly those of
the opcodes consumed.  Without a hint every next() is therefore one device step.
`encode_frame(target, is_aux, budget=K)` promises that K opcodes will be pulled;
they are then computed by one launch.  `Video.SPECULATE = N` (default: what one generator of movie.py's
pacing can be asked for -- 292 opcodes, the gap between two DHGR bank flips, or a frame's worth in HGR; 0 = one device
step per next()) gets batched launches without a promise: N opcodes are produced from a
device-side snapshot and rolled back + replayed if fewer were consumed.  Batched, many-stream encoding
(what bench.py measures) goes through stream_batch.StreamBatch instead.

Look-ahead (round 6, `Video.LOOKAHEAD`): behind a movie.py-paced DHGR generator whose opcodes end at a bank flip, the generator
the caller will start next -- the other bank of the SAME target (movie.py:139-148) -- is enqueued at once, on a second
snapshot, so that the device computes it while Python hands out this generator's opcodes.  If encode_frame() is then called
with that target and bank (and nobody touched the state or the global generators in between) its launch has already run; if not,
the snapshot is restored and nothing of it is observable.

Live hand-over (round 6, `Video.LIVE`): a speculative launch no longer has to END before its first opcode is handed out.  The
eight-wave team kernel writes every opcode, as one tagged 8-byte store, into a queue in coherent host memory
(include/iivision.h: iiv_encode_live), and the generator yields opcode i as soon as slot i carries the launch's tag -- while the
kernel works on i + 1.  One clip's chain of steps runs at ~0.55 us per opcode on the device, about what a Python caller takes
to pull one: the two now overlap instead of adding up.  What is observable is unchanged: a generator abandoned after k opcodes
is rolled back and replayed for exactly k, as before; a launch that ends short (one of the reference's asserts) marks the queue
behind its last opcode, and the generator then steps exactly from there, so the next() that raises in the reference raises here.
"""

import ctypes
import operator
import random
import time
from typing import Iterator, List, Tuple

import numpy as np

import _iiv_native as native
import screen
from palette import Palette
from video_mode import VideoMode

_np_global = [None, None]   # [the bit generator the address below belongs to, that address or 0 = "use get_state / set_state"]


def _np_rng_addr():
    """Address of the process-wide np.random MT19937 state -- key[624] (u32) then pos (i32), the words np.random.get_state()
    reports -- through numpy's own ctypes interface (BitGenerator.ctypes.state_address): reading and writing 2500 bytes there
    costs under a microsecond where get_state() / set_state() cost ~50 each, twice per generator.  The layout is numpy's
    private business (mt19937_state: uint32 key[624]; int pos), so the address is believed only after the bytes there have
    been seen to BE what get_state() reports, before and after a draw that moves the position; if they are not, 0 is
    returned and this module goes through get_state() / set_state() (slower, same results)."""
    bg = np.random.mtrand._rand._bit_generator
    if _np_global[0] is not bg:
        if type(bg).__name__ != "MT19937":
            raise RuntimeError("np.random's global generator is not the MT19937 the reference draws from")
        addr = 0
        try:
            cand = int(bg.ctypes.state_address)

            def same():
                _, key, pos = np.random.get_state()[:3]
                raw = np.frombuffer(ctypes.string_at(cand, 2500), dtype=np.uint32)
                return bool(np.array_equal(raw[:624], np.asarray(key, dtype=np.uint32)) and int(raw[624]) == int(pos))
            saved = np.random.get_state()
            ok = same()
            np.random.random_sample()           # (moves pos, or refills the block)
            ok = ok and same()
            np.random.set_state(saved)          # (the caller's stream is where it was)
            if ok and same():
                addr = cand
        except Exception:
            addr = 0
        _np_global[0], _np_global[1] = bg, addr
    return _np_global[1]


def _np_rng_raw():
    """np.random's MT19937 words as 2500 bytes: key[624], pos"""
    addr = _np_rng_addr()
    if addr:
        return ctypes.string_at(addr, 2500)
    _, key, pos = np.random.get_state()[:3]
    return np.asarray(key, dtype=np.uint32).tobytes() + np.array([pos], dtype=np.int32).tobytes()


def _np_rng_write(words):
    """the 625 words (ctypes array / buffer of 2500 bytes) become np.random's MT19937 state; has_gauss / cached_gaussian stay the caller's"""
    addr = _np_rng_addr()
    if addr:
        with np.random.mtrand._rand._bit_generator.lock:   # (the generator's own lock: nobody draws while the words change)
            ctypes.memmove(addr, words, 2500)
        return
    w = np.frombuffer(bytes(words), dtype=np.uint32)
    old = np.random.get_state()
    np.random.set_state((old[0], w[:624].copy(), int(w[624].astype(np.int32) if hasattr(w[624], "astype") else w[624]), old[3], old[4]))


_py_global = [None, None]   # [the random.Random instance the address belongs to, that address or 0 = "use getstate / setstate"]


def _py_rng_addr():
    """Address of `index` in the process-wide `random` generator's C struct (CPython _randommodule.c RandomObject: PyObject_HEAD,
    int index, uint32_t state[624]) -- random.getstate()[1] is state[0..623] followed by index.  getstate() builds a tuple of 625
    ints and setstate() parses one (~15 and ~40 microseconds, both at every generator start); the 2500 bytes themselves move in
    under one.  The layout is CPython's private business, so -- as for np.random above -- the address is believed only after
    the bytes there have been seen to BE what getstate() reports, before and after a draw and after a write through it; if
    not, 0 is returned and this module goes through getstate() / setstate() (slower, same results)."""
    inst = getattr(random, "_inst", None)
    if _py_global[0] is not inst or inst is None:
        addr = 0
        try:
            if inst is not None and random.getstate.__self__ is inst and type(inst).__mro__[1].__name__ == "Random":
                cand = id(inst) + object.__basicsize__      # (PyObject_HEAD of a non-GC base: refcount, type)

                def same():
                    words = np.array(random.getstate()[1], dtype=np.uint32)
                    raw = np.frombuffer(ctypes.string_at(cand, 2500), dtype=np.uint32)
                    return bool(np.array_equal(raw[1:], words[:624]) and raw[0] == words[624])
                saved = random.getstate()
                try:
                    ok = same()
                    random.getrandbits(8)               # (moves index, or refills the block)
                    ok = ok and same()
                    if ok:
                        probe = np.arange(7, 7 + 625, dtype=np.uint32)
                        probe[0] = 3                    # index
                        ctypes.memmove(cand, probe.ctypes.data, 2500)
                        got = random.getstate()[1]
                        ok = got[624] == 3 and got[:624] == tuple(range(8, 8 + 624))
                finally:
                    random.setstate(saved)              # (the caller's stream is where it was)
                if ok and same():
                    addr = cand
        except Exception:
            addr = 0
        _py_global[0], _py_global[1] = inst, addr
    return _py_global[1]


def _py_rng_raw():
    """random's MT19937 words as 2500 bytes in getstate() order: state[624], index"""
    addr = _py_rng_addr()
    if addr:
        return ctypes.string_at(addr + 4, 2496) + ctypes.string_at(addr, 4)
    return np.array(random.getstate()[1], dtype=np.uint32).tobytes()


def _py_rng_write(words):
    """the 625 words (getstate() order; a ctypes array / buffer of 2500 bytes) become random's state; gauss_next is cleared, as
    random.setstate((3, words, None)) would"""
    addr = _py_rng_addr()
    if addr:
        base = ctypes.addressof(words) if isinstance(words, ctypes.Array) else np.frombuffer(words, dtype=np.uint8).ctypes.data
        ctypes.memmove(addr + 4, base, 2496)
        ctypes.memmove(addr, base + 2496, 4)
        random._inst.gauss_next = None
        return
    random.setstate((3, tuple(np.frombuffer(bytes(words), dtype=np.uint32).tolist()), None))


class _Chunk:
    """A speculative launch whose opcodes are being handed out: what _settle needs to make the device state that of the
    opcodes CONSUMED.  The hand-out is `yield from` a list iterator (no Python statement per opcode), so the count is read
    off that iterator when somebody asks: opcodes given to iterators so far, less what the current one still holds."""
    __slots__ = ("token", "restart", "produced", "prev_live", "slot", "items", "it", "base")

    def __init__(self, token, restart, produced, prev_live, slot):
        self.token, self.restart, self.produced, self.prev_live, self.slot = token, restart, produced, prev_live, slot
        self.items, self.it, self.base = None, None, 0

    def hand_out(self, items):
        """the iterator to `yield from`: the next opcodes of the launch"""
        self.items, self.base = items, self.base + len(items)
        self.it = iter(items)
        return self.it

    def consumed(self):
        return self.base - (operator.length_hint(self.it) if self.it is not None else 0)

    def stop(self):
        """nothing more is handed out (the launch was settled underneath its generator)"""
        if self.items is not None:
            del self.items[:]


class Video:
    """Encodes sequence of images into prioritized screen byte changes."""

    CLOCK_SPEED = 1024 * 1024  # type: int

    #: Opcodes produced per device call when encode_frame() got no `budget`.
    #: 0 or 1 = one exact step per next().  N > 1 = speculate: N opcodes are produced
    #: from a device-side snapshot; if the generator is abandoned (or any state
    #: attribute is read) after k < N of them were consumed, the snapshot is restored
    #: and exactly k are replayed, so observable state is always that of the consumed
    #: opcodes.  Nothing observable depends on it (an assertion of the reference that would
    #: fire inside the unconsumed part of a chunk makes the generator fall back to exact
    #: stepping), only the speed does.
    #: None (default): the number of opcodes movie.py's pacing will pull from this generator, as far as this object can tell
    #: from what the caller has shown it (_paced_chunk): tick() tells it the tick count, so the tick that starts the next
    #: frame is known (video.py:64-70); a generator started for the other bank without a new frame was a bank flip, and the
    #: next one comes 292 opcodes later (2044 bytes of a 2 KiB socket frame / 7, movie.py:139-148).  A right guess is one
    #: launch and nothing to roll back; a wrong one costs a roll-back or a second launch, never a different opcode.
    #: Without tick() calls: DHGR 292, HGR a frame's worth (ticks_per_frame).
    SPECULATE = None

    #: True: after every next() the host arrays and the *global* random / np.random states
    #: are those of the reference at that point (one full state round trip per opcode).
    #: False (default): the arrays are synchronised whenever a state attribute is read, the global
    #: RNG positions and out_of_work also whenever another generator starts (a 5 KB round trip:
    #: iiv_encoder_get_video_brief); draws from / reseeds of random or np.random between two
    #: generators are noticed and carried to the device.
    STRICT_SYNC = False

    #: True (default): run one generator ahead of a movie.py-paced caller (module docstring).  Speed only.
    LOOKAHEAD = True

    #: True (default): hand the opcodes of a speculative launch out WHILE the kernel produces them (module docstring: live
    #: hand-over) instead of after it has ended.  Speed only; encoders whose options keep them off the team kernel
    #: (joint_content) fall back by themselves.
    LIVE = True
    LIVE_TIMEOUT = 20.0   # seconds without a new opcode before the launch is declared dead

    def __init__(
            self,
            frame_grabber,
            ticks_per_second: float,
            mode: VideoMode = VideoMode.HGR,
            palette: Palette = Palette.NTSC,
            joint_content: bool = False,
            fourth_offset: bool = False
    ):
        """joint_content (not in the reference's signature, default off): choose every opcode's
        content byte jointly with its extra offsets -- the "global optimization" of the reference's
        README.md:212-215 (include/iivision.h: IIV_CONTENT_JOINT).  Less error per opcode, NOT the
        reference's opcode stream.
        fourth_offset (likewise not in the reference's signature, default off): up to three extra offsets per
        opcode -- the "3 more offsets" of video.py:146 -- instead of the reference's two and a copy of the first
        (video.py:180-186; IIV_OPT_FOURTH_OFFSET).  NOT the reference's opcode stream."""
        self.mode = mode  # type: VideoMode
        self.frame_grabber = frame_grabber
        self.ticks_per_second = float(ticks_per_second)  # type: float
        self.ticks_per_frame = (
                self.ticks_per_second / frame_grabber.input_frame_rate
        )  # type: float
        self.frame_number = 0  # type: int
        self.palette = palette  # type: Palette
        self._pending = None  # speculative chunk not yet fully consumed
        self._host_current = True  # host arrays / global RNG states equal the device's
        self._touched = True  # a state attribute was handed out since the last upload

        # Empty screen (video.py:37-53); the pixelmap aliases the memory maps
        self._memory_map = screen.MemoryMap(screen_page=1)
        self._aux_memory_map = None
        if self.mode == VideoMode.DHGR:
            self._aux_memory_map = screen.MemoryMap(screen_page=1)
            self._pixelmap = screen.DHGRBitmap(
                palette=palette, main_memory=self._memory_map, aux_memory=self._aux_memory_map)
        else:
            self._pixelmap = screen.HGRBitmap(palette=palette, main_memory=self._memory_map)

        # Pending edit weights, accumulated across frames (video.py:55-58)
        self._update_priority = np.zeros((32, 256), dtype=np.int32)
        self._aux_update_priority = None
        if self.mode == VideoMode.DHGR:
            self._aux_update_priority = np.zeros((32, 256), dtype=np.int32)

        # True once the main / aux bank has run out of work (video.py:60-62)
        self._out_of_work = {True: False, False: False}

        tables = self._pixelmap.edit_distances(palette)
        self._mode_id = native.DHGR if mode == VideoMode.DHGR else native.HGR
        self._enc = native.Encoder(self._mode_id, tables.table, tables.store, n_streams=1, dm=tables.dm)
        if joint_content:
            self._enc.set_content_choice(True)
        if fourth_offset:
            self._enc.set_fourth_offset(True)
        self._live = None  # the generator whose state the device currently holds
        self._vs = native.VideoState()   # one staging buffer for every state round trip
        self._vb_mem = None              # the brief lives in pinned memory: it is also fetched behind a launch, asynchronously
        self._vb = native.VideoBrief()
        self._brief_applied = True       # what self._vb says about the global RNG positions / out_of_work has reached the host
        self._rng_seen = None  # the global (random, np.random) states as this object last left or read them
        self._brief_fresh = False  # self._vb describes the device state as it stands
        self._dev_main = self._dev_aux = None   # the live generator's target on the device
        self._up_main = self._up_aux = None     # ... and what was last copied there (a bank flip within a frame re-uses it)
        self._ops_dev = self._ops_host = None   # opcode buffers: device, and pinned host memory the launch copies into
        # what the caller's pacing has shown so far (only the size of speculative launches depends on it)
        self._tick_now = None     # the latest tick() argument
        self._ops_done = 0        # opcodes consumed from settled chunks
        self._flip_base = -1      # _ops_done at the last bank flip seen (movie.py's first socket frame holds 291 opcodes)
        self._last_bank = None    # is_aux of the latest generator that ran
        # the generator enqueued ahead of the caller (LOOKAHEAD): None, or what it was launched for and where its results land
        self._ahead = None
        self.lookahead_stats = {"launched": 0, "adopted": 0, "undone": 0}   # (what became of the generators enqueued ahead)
        self._ahead_bufs = None   # second set of opcode / brief buffers (the look-ahead's results must not overwrite the live generator's)
        # live hand-over: the two host queues the team kernel writes opcodes into (None until first used; False: this encoder's
        # launches do not run that kernel), the tag of the latest launch, the event behind the brief that follows a launch
        self._live_q = None
        self._live_tag = 0
        self._live_epoch = 0             # how often the tag has started over; per queue: the epoch it was last cleared in
        self._live_q_epoch = [0, 0]
        self._brief_event = None
        # (launches handed out live; polls of the queue; polls that found nothing and waited, and for how long)
        self.live_stats = {"launches": 0, "takes": 0, "waits": 0, "wait_s": 0.0, "first_wait_s": 0.0}

    # ---- the reference's public attributes; reading one settles any speculation first
    def _settled(name):  # noqa: N805
        def get(self):
            self._settle()
            self._touched = True  # the caller may change what it gets
            return getattr(self, name)

        def set_(self, value):
            self._settle()
            self._touched = True
            setattr(self, name, value)
        return property(get, set_)

    memory_map = _settled("_memory_map")
    pixelmap = _settled("_pixelmap")
    update_priority = _settled("_update_priority")

    @property
    def out_of_work(self):
        self._settle()
        self._touched = True
        return self._out_of_work

    @out_of_work.setter
    def out_of_work(self, value):
        # movie.py:96 assigns a fresh dict at every frame: only the two flags travel, not the whole state
        self._settle(download=False)
        self._out_of_work = value
        if not self._touched:  # the device's copy is the one the next launch reads
            # (enqueued behind whatever is in flight: the next launch, on the same stream, reads them)
            self._enc.set_state_async(native.STATE_OUT_OF_WORK, np.array([int(bool(value[False])), int(bool(value[True]))], np.int32))
            # (the brief at hand stays good: the two flags are all that changed on the device, and they are known -- once it HAS
            # arrived: a brief still on its way behind a live launch would land on top of what is written here)
            if self._brief_event is not None:
                self._brief_event.synchronize()
                self._brief_event = None
            self._vb.out_of_work[0] = int(bool(value[False]))
            self._vb.out_of_work[1] = int(bool(value[True]))

    @property
    def aux_memory_map(self):
        if self._aux_memory_map is None:
            raise AttributeError("aux_memory_map")  # HGR Video has none (video.py:40-42)
        self._settle()
        self._touched = True
        return self._aux_memory_map

    @property
    def aux_update_priority(self):
        if self._aux_update_priority is None:
            raise AttributeError("aux_update_priority")
        self._settle()
        self._touched = True
        return self._aux_update_priority

    del _settled

    def tick(self, ticks: int) -> bool:
        """Keep track of when it is time for a new image frame (video.py:64-70)."""
        self._tick_now = ticks
        if ticks >= (self.ticks_per_frame * self.frame_number):
            self.frame_number += 1
            return True
        return False

    def _paced_chunk(self, after=0, flipped=False, why=None):
        """How many opcodes a movie.py-paced caller will pull before it starts another generator (SPECULATE = None).
        after / flipped: the same question for the generator BEHIND the next `after` opcodes (flipped: it starts at a bank
        flip) -- what the look-ahead is sized by.  why (a list): receives "flip" if the bank flip is what ends the count, "frame"
        if the next frame does, None if neither is known."""
        dhgr = self.mode == VideoMode.DHGR
        n = 292 if dhgr else max(1, int(round(self.ticks_per_frame)))
        reason = None
        if self._tick_now is not None:
            # pulls at ticks _tick_now, _tick_now + 1, ... up to the tick in front of the one that starts a frame
            to_frame = int(-(-(self.ticks_per_frame * self.frame_number) // 1)) - (int(self._tick_now) + after)
            if to_frame >= 1:
                if not dhgr or to_frame < n:
                    reason = "frame"
                n = to_frame if not dhgr else min(n, to_frame)
            elif after:
                n, reason = 0, "frame"      # (the frame ends with the opcodes in front: no generator of this frame follows)
        if dhgr and n > 0:
            to_flip = 292 - ((self._ops_done + after) - (self._ops_done + after if flipped else self._flip_base))
            if 1 <= to_flip <= n:
                if to_flip < n or reason is None:
                    reason = "flip"
                n = to_flip
        if why is not None:
            why.append(reason)
        return min(n, 2048)

    # ------------------------------------------------------------------ device sync

    def _upload(self):
        """Host state -> device, one call (iiv_encoder_set_video_state)."""
        st = self._vs
        dhgr = self.mode == VideoMode.DHGR
        st.array("mem_main", np.uint8, (32, 256))[...] = self._memory_map.page_offset
        st.array("up_main", np.int32, (32, 256))[...] = self._update_priority
        if dhgr:
            st.array("mem_aux", np.uint8, (32, 256))[...] = self._aux_memory_map.page_offset
            st.array("up_aux", np.int32, (32, 256))[...] = self._aux_update_priority
        py = _py_rng_raw()
        st.array("rng_py", np.uint32, (625,))[...] = np.frombuffer(py, dtype=np.uint32)
        raw = _np_rng_raw()
        st.array("rng_np", np.uint32, (625,))[...] = np.frombuffer(raw, dtype=np.uint32)
        self._rng_seen = (py, raw)
        # movie.py:96 resets the flags at every frame
        st.out_of_work[0] = int(bool(self._out_of_work[False]))
        st.out_of_work[1] = int(bool(self._out_of_work[True]))
        self._enc.set_video_state(st)
        self._touched = False
        self._brief_fresh = False

    def _download(self):
        """Device state -> host, one call; in place: callers (and self.pixelmap) hold references to the arrays."""
        st = self._enc.get_video_state(out=self._vs)
        self._memory_map.page_offset[...] = st.array("mem_main", np.uint8, (32, 256))
        self._update_priority[...] = st.array("up_main", np.int32, (32, 256))
        if self.mode == VideoMode.DHGR:
            self._aux_memory_map.page_offset[...] = st.array("mem_aux", np.uint8, (32, 256))
            self._aux_update_priority[...] = st.array("up_aux", np.int32, (32, 256))
        self._pixelmap.packed[...] = st.array("packed", np.uint64, (32, 128))
        # the device's RNG positions become the process's -- unless the caller has drawn from / reseeded random or
        # np.random since this object last synchronised them: then the caller's state stands (it is what the reference's
        # next generator would read, video.py:178,265,291) and travels to the device with the next launch
        if not self._global_rng_moved():
            self._set_global_rng(st)
        self._out_of_work[False] = bool(st.out_of_work[0])  # video.py:189
        self._out_of_work[True] = bool(st.out_of_work[1])
        self._host_current = True

    def _set_global_rng(self, st):
        """the device's random / np.random positions (st.rng_py, st.rng_np) become the process's"""
        _py_rng_write(st.rng_py)
        _np_rng_write(st.rng_np)
        self._rng_seen = (bytes(st.rng_py), bytes(st.rng_np))

    def _global_rng_moved(self):
        """did anyone draw from / reseed random or np.random since this object last synchronised them?"""
        if self._rng_seen is None:
            return True
        return (_py_rng_raw(), _np_rng_raw()) != self._rng_seen

    def _upload_rng(self):
        pyraw = _py_rng_raw()
        py = np.frombuffer(pyraw, dtype=np.uint32).copy()
        raw = _np_rng_raw()
        rn = np.frombuffer(raw, dtype=np.uint32).copy()
        self._enc.set_state_async(native.STATE_RNG_PY, py)
        self._enc.set_state_async(native.STATE_RNG_NP, rn)
        self._brief_fresh = False
        self._rng_seen = (pyraw, raw)

    def _pinned_brief(self):
        """self._vb in page-locked memory (allocated with the first launch: torch is needed for it)"""
        if self._vb_mem is None:
            import ctypes
            import torch
            self._vb_mem = torch.empty(ctypes.sizeof(native.VideoBrief), dtype=torch.uint8).pin_memory()
            self._vb = native.VideoBrief.from_address(self._vb_mem.data_ptr())
        return self._vb

    def _sync_brief(self):
        """Settle the device state and bring home the small things: global RNG positions, out_of_work
        (and the numbers encode_frame prints / asserts).  The arrays stay on the device.  The brief itself usually is at
        hand already: it travels behind every launch (_launch), valid as long as the launch's opcodes are all consumed."""
        self._settle(download=False, keep_ahead=self._brief_fresh)   # (a brief must be fetched: the device must stand where the caller is)
        if self._host_current:
            return None  # nothing on the device is newer than what the host holds
        if not self._brief_fresh:
            self._enc.get_video_brief(out=self._pinned_brief())
            self._brief_fresh = True
            self._brief_applied = False
        b = self._vb
        if self._brief_event is not None:
            # (a live launch: its opcodes were handed out while it ran; the brief behind it arrives with its end)
            self._brief_event.synchronize()
            self._brief_event = None
        if not self._brief_applied:
            if not self._global_rng_moved():  # (else the caller's draws / reseed win: uploaded at the next launch)
                self._set_global_rng(b)
            self._out_of_work[False] = bool(b.out_of_work[0])
            self._out_of_work[True] = bool(b.out_of_work[1])
            self._brief_applied = True
        return b

    def _launch_buffers(self, n_ops):
        import torch
        if self._ops_dev is None or self._ops_dev.shape[1] < n_ops:
            cap = max(n_ops, 2048)
            self._ops_dev = torch.empty((1, cap, 6), dtype=torch.uint8, device="cuda")
            self._ops_host = torch.empty((cap, 6), dtype=torch.uint8).pin_memory()

    def _launch(self, token, restart, n_ops, fetch=True):
        """[prologue +] n_ops greedy steps on the device state as it stands.  fetch=False: a replay of opcodes the caller has
        already consumed (after a roll-back): nothing to bring home and nothing that can fail -- the speculative launch they
        came from passed its check, and this is a prefix of it -- so the launch is only enqueued."""
        import torch
        self._brief_fresh = False
        self._brief_event = None
        n_ops = int(n_ops)
        self._launch_buffers(n_ops)
        ops = self._enc.encode(token.fm, token.fa, [(0, int(bool(token.is_aux)), int(restart), n_ops)], ops_out=self._ops_dev)
        self._host_current = False
        if not fetch:
            return None
        # the opcodes and the brief of the state behind them ride home on the stream, one wait for everything: if the caller
        # pulls all of them -- the rule when the launch was sized by its pacing -- the next generator starts without
        # another round trip
        self._ops_host[:n_ops].copy_(ops[0], non_blocking=True)
        self._enc.get_video_brief_async(self._pinned_brief())
        self._enc.check()
        self._brief_fresh = True
        self._brief_applied = False
        return self._ops_host[:n_ops].numpy()

    def _live_queues(self):
        """the two host queues of the live hand-over, or None if this Video does without (LIVE off, or its encoder's
        launches do not run the team kernel: found out at the first attempt)"""
        if not self.LIVE or self._live_q is False or self.STRICT_SYNC:
            return None
        if self._live_q is None:
            try:
                self._live_q = [self._enc.live_queue(0), self._enc.live_queue(1)]
            except (native.IIVError, AttributeError):
                self._live_q = False
                return None
        return self._live_q

    def _launch_live(self, token, is_aux, restart, n_ops, qslot, ops_dev, vb):
        """[prologue +] n_ops greedy steps, the opcodes appearing one by one in host queue `qslot` under a fresh tag; the brief
        of the state behind them follows into `vb` (pinned), an event behind it.  Nothing waits.  Returns what the consumer
        needs (_live_take), or None -- nothing launched -- if this encoder cannot (the caller then launches the old way)."""
        import torch
        qs = self._live_queues()
        if qs is None or n_ops > len(qs[qslot]):
            return None
        if self._live_tag >= 65535:
            self._live_tag = 0
            self._live_epoch += 1
        self._live_tag += 1
        if self._live_q_epoch[qslot] != self._live_epoch:
            # the 16-bit tag has started over since this queue was last cleared: a slot that no launch since has written could
            # still carry a tag that is about to be used again.  Wait for whatever may still write into the queue and clear
            # it -- THIS queue only: the other one may hold opcodes of the live generator that are yet to be handed out (this
            # launch is then the one enqueued ahead of it); its turn comes with its own next launch
            torch.cuda.current_stream().synchronize()
            qs[qslot][:] = 0
            self._live_q_epoch[qslot] = self._live_epoch
        try:
            self._enc.encode_live(token.fm, token.fa, (0, int(bool(is_aux)), int(restart), int(n_ops)), ops_dev, qslot, self._live_tag)
        except native.IIVError as e:
            if e.code != native.ERR_INVALID:
                raise
            self._live_q = False     # (refused before anything was launched: options that keep the encoder off the team kernel)
            return None
        self._host_current = False
        self._enc.get_video_brief_async(vb)
        ev = torch.cuda.Event()
        ev.record()
        self.live_stats["launches"] += 1
        q = qs[qslot]
        # (views of the queue: the tags as uint16 -- the top quarter of every slot --, the slots as bytes)
        return dict(q=q, q16=q.view(np.uint16).reshape(-1, 4)[:, 3], q8=q.view(np.uint8).reshape(-1, 8), tag=self._live_tag, n=int(n_ops), event=ev)

    def _live_take(self, lv, k):
        """Opcodes k .. of a live launch that have arrived -- at least one: this waits for slot k -- as a list of
        (page, content, offsets) tuples, and whether the launch ENDED behind them, short of its n_ops (then the list may be
        empty)."""
        q, tag, n = lv["q"], lv["tag"], lv["n"]
        st = self.live_stats
        st["takes"] += 1
        if (int(q[k]) >> 48) != tag:
            t0 = time.perf_counter()
            spins = 0
            while (int(q[k]) >> 48) != tag:
                spins += 1
                if not spins & 0xfff and time.perf_counter() - t0 > self.LIVE_TIMEOUT:
                    self._enc.check()     # (synchronises; raises what the device reports)
                    if (int(q[k]) >> 48) != tag:
                        raise RuntimeError("live hand-over: opcode %d of %d never arrived" % (k, n))
            dt = time.perf_counter() - t0
            st["waits"] += 1
            st["wait_s"] += dt
            if k == 0:
                st["first_wait_s"] += dt     # (of that: for a launch's first opcode -- its prologue, mostly)
        # the slots from k on that carry the tag, up to the first that does not (the waves of a round commit in any order)
        ok = lv["q16"][k:n] == tag
        r = int(ok.argmin()) or len(ok)      # (slot k carries it: argmin is 0 only when all of them do)
        b = lv["q8"][k:k + r]
        # an end mark is the last slot its launch writes: if it has arrived it is the last of these
        ended = int(b[r - 1, 0]) == 0xFF
        if ended:
            b = b[:r - 1]
        return list(zip(b[:, 0].tolist(), b[:, 1].tolist(), b[:, 2:6].tolist())), ended

    def _look_ahead(self, token, n_live, slot):
        """Behind the live generator's launch (its n_live opcodes are in hand): if they end at a bank flip inside the frame,
        enqueue the generator movie.py:139-148 starts next -- the other bank, the same target -- on snapshot `slot`, with the
        opcodes and the brief of the state behind them copied to a second set of pinned buffers.  Nothing waits."""
        import torch
        why = []
        self._paced_chunk(after=0, why=why)        # what ends the live generator's count as the pacing stands
        # (the live chunk was sized by this very call before its launch; only a count that a bank flip ended has a successor
        # inside the frame, and only if the caller's pacing is known)
        if why[0] != "flip" or self._tick_now is None:
            return
        n = self._paced_chunk(after=n_live, flipped=True)
        if n < 1:
            return
        if self._ahead_bufs is None:
            self._ahead_bufs = dict(
                ops_dev=torch.empty((1, 2048, 6), dtype=torch.uint8, device="cuda"),
                ops_host=torch.empty((2048, 6), dtype=torch.uint8).pin_memory(),
                vb_mem=torch.empty(ctypes.sizeof(native.VideoBrief), dtype=torch.uint8).pin_memory())
        bufs = self._ahead_bufs
        self._enc.snapshot(slot)
        vb = native.VideoBrief.from_address(bufs["vb_mem"].data_ptr())
        lv = self._launch_live(token, not token.is_aux, 1, int(n), slot, bufs["ops_dev"], vb)
        if lv is None:
            ops = self._enc.encode(token.fm, token.fa, [(0, int(not token.is_aux), 1, int(n))], ops_out=bufs["ops_dev"])
            bufs["ops_host"][:n].copy_(ops[0], non_blocking=True)
            self._enc.get_video_brief_async(vb)
        self._host_current = False
        self._ahead = dict(is_aux=not token.is_aux, main=token.main, aux=token.aux, n=int(n), slot=slot, live=lv)
        self.lookahead_stats["launched"] += 1

    def _adopt(self, a):
        """The generator enqueued ahead is the one the caller asked for: its launch has run (or is running); wait, check, and
        swap the buffer sets so that the brief and the opcodes of this generator are the current ones."""
        lv = a.get("live")
        if lv is None:
            self._enc.check()      # (synchronises; raises what the reference's asserts would)
        self.lookahead_stats["adopted"] += 1
        bufs = self._ahead_bufs
        self._pinned_brief()
        bufs["ops_host"], self._ops_host = self._ops_host, bufs["ops_host"]
        bufs["ops_dev"], self._ops_dev = self._ops_dev, bufs["ops_dev"]
        bufs["vb_mem"], self._vb_mem = self._vb_mem, bufs["vb_mem"]
        self._vb = native.VideoBrief.from_address(self._vb_mem.data_ptr())
        self._host_current = False
        self._brief_fresh = True
        self._brief_applied = False
        self._brief_event = lv["event"] if lv is not None else None
        return (None, lv) if lv is not None else (self._ops_host[:a["n"]].numpy(), None)

    def _settle(self, download=True, keep_ahead=False):
        """Make the device state -- and, with download, the host's -- reflect exactly the opcodes consumed so far.
        keep_ahead: a generator enqueued ahead of the caller (LOOKAHEAD) stays in flight if the live generator's opcodes were
        all consumed (the caller is about to ask for a generator: maybe this one); otherwise it is undone here."""
        p = self._pending
        self._pending = None
        consumed = 0
        if p is not None:
            consumed = p.consumed()
            p.stop()
            self._ops_done += consumed
        partial = p is not None and consumed < p.produced
        a = self._ahead
        if a is not None and (partial or download or not keep_ahead):
            self._ahead = None
            self.lookahead_stats["undone"] += 1
            if not partial:
                # back to the state behind the live generator's opcodes (what self._vb describes, if it is fresh)
                self._enc.rollback(a["slot"])
                self._host_current = False
        if partial:
            # abandoned mid-chunk: restore the snapshot and replay only what was consumed (a look-ahead behind it goes with it)
            self._enc.rollback(p.slot)
            self._host_current = False
            self._brief_fresh = False   # (what travelled behind the launch describes all of its opcodes)
            if consumed:
                self._launch(p.token, p.restart, consumed, fetch=False)
            elif p.restart:
                self._live = p.prev_live  # the prologue never happened
                p.token.started = False
        if download and not self._host_current:
            self._download()

    # ------------------------------------------------------------------ encode

    def encode_frame(
            self,
            target: screen.Bitmap,
            is_aux: bool,
            budget: int = None,
    ) -> Iterator[Tuple[int, int, List[int]]]:
        """Converge towards target frame in priority order of edit distance.

        Lazy generator, as in the reference (video.py:72-93): nothing happens until
        the first next(); it never terminates (pads forever once out of work).
        """
        if is_aux and self._aux_memory_map is None:
            raise AttributeError("aux_memory_map")  # as the reference's HGR Video (video.py:79-80)
        self._settle(download=False, keep_ahead=True)
        if self._host_current:
            memory_map = self._aux_memory_map if is_aux else self._memory_map
            update_priority = self._aux_update_priority if is_aux else self._update_priority
            # Make sure nothing is leaking into screen holes (video.py:87)
            assert np.count_nonzero(memory_map.page_offset[screen.SCREEN_HOLES]) == 0
            similarity = update_priority.mean()
        else:
            # the state lives on the device: fetch what these lines look at (and what the caller could
            # observe between two generators: the global RNG positions, out_of_work) -- 5 KB, not 300
            b = self._sync_brief()
            assert b.hole_bytes[1 if is_aux else 0] == 0
            similarity = b.priority_sum[1 if is_aux else 0] / 8192.0  # == update_priority.mean(), exactly

        print("Similarity %f" % similarity)

        yield from self._index_changes(target, is_aux, budget)

    def _index_changes(self, target_pixelmap, is_aux, budget):
        import torch

        class _Token:
            started = False

        token = _Token()
        token.is_aux = bool(is_aux)
        # the target lives in ONE pair of device buffers per Video (only the latest generator can run,
        # see below); this generator's copy goes there when it starts, after the previous one is settled
        token.main = np.ascontiguousarray(target_pixelmap.main_memory.page_offset, dtype=np.uint8).reshape(1, 1, 32, 256).copy()
        token.aux = None
        if self.mode == VideoMode.DHGR:
            token.aux = np.ascontiguousarray(target_pixelmap.aux_memory.page_offset, dtype=np.uint8).reshape(1, 1, 32, 256).copy()
        if self._dev_main is None:
            self._dev_main = torch.empty((1, 1, 32, 256), dtype=torch.uint8, device="cuda")
            self._dev_aux = torch.empty((1, 1, 32, 256), dtype=torch.uint8, device="cuda") if token.aux is not None else None
        token.fm, token.fa = self._dev_main, self._dev_aux
        spec = self.SPECULATE
        paced = spec is None and not budget and not self.STRICT_SYNC
        chunk = int(budget) if budget else 1 if self.STRICT_SYNC else max(1, int(spec)) if not paced else 0
        speculative = paced or (not budget and chunk > 1)
        try:
            while True:
                restart = 0 if self._live is token else 1
                if restart and token.started:
                    raise RuntimeError("this encode_frame() generator cannot be resumed: another generator "
                                       "has run on this Video since (the reference's heap is not kept)")
                prev_live = self._live
                adopt = None
                if restart and self._ahead is not None:
                    # is this the generator that was enqueued ahead?  Same bank, same target bytes, the live generator's
                    # opcodes all consumed, nothing touched, nobody drew from the global generators since
                    a = self._ahead
                    ok = (paced and not self._touched and not self.STRICT_SYNC and a["is_aux"] == token.is_aux
                          and (self._pending is None or self._pending.consumed() == self._pending.produced)
                          and np.array_equal(a["main"], token.main)
                          and (token.aux is None or np.array_equal(a["aux"], token.aux)) and not self._global_rng_moved())
                    self._settle(download=False, keep_ahead=ok)
                    if ok and self._ahead is a:
                        adopt, self._ahead = a, None
                elif restart or self._pending is not None or self._ahead is not None:
                    self._settle(download=False)  # the previous generator's speculation ends here
                if self._touched or self.STRICT_SYNC:
                    self._settle()
                    self._upload()
                elif restart and self._global_rng_moved():
                    self._upload_rng()  # someone drew from / reseeded random or np.random in between
                if restart:
                    if self._last_bank is not None and self._last_bank != token.is_aux:
                        self._flip_base = self._ops_done  # (the previous generator is settled: _ops_done is exact)
                    self._last_bank = token.is_aux
                if paced and speculative:
                    chunk = max(1, self._paced_chunk())
                if restart:
                    # (the generators of one frame share their target: only what differs from the last upload travels)
                    if self._up_main is None or not np.array_equal(self._up_main, token.main):
                        self._dev_main.copy_(torch.from_numpy(token.main))
                        self._up_main = token.main
                    if token.aux is not None and (self._up_aux is None or not np.array_equal(self._up_aux, token.aux)):
                        self._dev_aux.copy_(torch.from_numpy(token.aux))
                        self._up_aux = token.aux
                slot = 0
                live = None     # the launch's opcodes arrive one by one in a host queue (LIVE): what _live_take needs
                if speculative:
                    try:
                        if adopt is not None:
                            slot = adopt["slot"]
                            ops, live = self._adopt(adopt)
                            chunk = adopt["n"]
                        else:
                            self._enc.snapshot(slot)
                            self._brief_fresh = False
                            self._brief_event = None
                            self._launch_buffers(chunk)
                            live = self._launch_live(token, token.is_aux, restart, chunk, slot, self._ops_dev, self._pinned_brief())
                            if live is not None:
                                self._brief_fresh = True
                                self._brief_applied = False
                                self._brief_event = live["event"]
                            else:
                                ops = self._launch(token, restart, chunk)
                    except native.IIVAssertionError:
                        # one of the reference's asserts fires somewhere in this chunk -- maybe past
                        # what the caller will pull: step exactly from here on, so that it is raised
                        # by the next() that would raise it in the reference
                        self._enc.rollback(slot)
                        self._host_current = False
                        self._brief_fresh = False
                        speculative, chunk = False, 1
                        continue
                elif budget and chunk >= 32 and not self.STRICT_SYNC:
                    # a promised budget: one launch, nothing to roll back -- handed out while it runs, too
                    self._brief_fresh = False
                    self._brief_event = None
                    self._launch_buffers(chunk)
                    live = self._launch_live(token, token.is_aux, restart, chunk, 0, self._ops_dev, self._pinned_brief())
                    if live is not None:
                        self._brief_fresh = True
                        self._brief_applied = False
                        self._brief_event = live["event"]
                    else:
                        ops = self._launch(token, restart, chunk)
                else:
                    ops = self._launch(token, restart, chunk)
                self._live = token
                token.started = True
                if self.STRICT_SYNC:
                    self._download()
                rec = None
                produced = live["n"] if live is not None else len(ops)
                if speculative:
                    rec = _Chunk(token, restart, produced, prev_live, slot)
                    self._pending = rec
                    if paced and self.LOOKAHEAD and self.mode == VideoMode.DHGR:
                        self._look_ahead(token, produced, 1 - slot)
                if live is not None and rec is None:
                    # (a promised budget, live: nothing is pending -- the state may run ahead of the caller, that is the promise)
                    tally = _Chunk(token, restart, produced, prev_live, slot)
                    ended = False
                    while tally.base < produced and not ended:
                        items, ended = self._live_take(live, tally.base)
                        yield from tally.hand_out(items)
                    if tally.base < produced:
                        self._enc.check()     # the launch ended short: raises what the device reports, at the next() it belongs to
                        raise RuntimeError("live hand-over: the launch ended after %d of %d opcodes" % (tally.base, produced))
                    self._ops_done += produced
                    chunk = 1
                    continue
                if live is not None:
                    # hand out what has arrived, as it arrives
                    ended = False
                    while rec.base < produced and not ended and self._pending is rec:
                        items, ended = self._live_take(live, rec.base)
                        yield from rec.hand_out(items)
                    if self._pending is not rec:
                        continue   # settled underneath us: the rest of this launch was rolled back
                    if rec.base < produced:
                        # the launch ended short of its opcodes: one of the reference's asserts fires at the next one (or an
                        # internal limit was hit).  Settle to exactly the opcodes consumed -- roll back, replay them -- and
                        # step exactly from here: the next() that raises in the reference raises here
                        self._settle(download=False)
                        speculative, chunk = False, 1
                        continue
                    self._pending = None
                    self._ops_done += produced
                    continue
                # plain ints, converted once per chunk
                items = list(zip(ops[:, 0].tolist(), ops[:, 1].tolist(), ops[:, 2:6].tolist()))
                if rec is None:
                    yield from items
                    self._ops_done += len(items)
                else:
                    yield from rec.hand_out(items)
                    if self._pending is rec:   # (else: settled underneath us, the rest of this chunk was rolled back)
                        self._pending = None
                        self._ops_done += len(items)
                if not speculative:
                    chunk = 1
        except GeneratorExit:
            # abandoned (movie.py:94-101 rebinds op_seq: CPython finalises the old generator right there):
            # undo what was speculated beyond the consumed opcodes and bring the global RNG positions
            # home now, so that draws made before the next generator starts continue the right stream
            if self._live is token:
                try:
                    self._sync_brief()
                except Exception:
                    pass  # (interpreter shutdown, closed encoder: nothing left to keep consistent)
            raise
