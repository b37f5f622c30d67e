// Golden-vector generator.  Runs in THIS container only (node 12, --experimental-modules), from a
// scratch directory that holds .mjs copies of the reference's legacy JS codec
// (/root/reference/javascript/test/legacy/{encoding,common,columnar}.js) — see tools/golden/README.md
// for the recipe.  It imports the reference's own second implementation of the codecs and writes
// plain-data vectors (inputs + expected outputs) that are committed under tests/golden/.
// Nothing from the reference is copied into the repository; only the data this script prints.
import { Encoder, Decoder, RLEEncoder, RLEDecoder, DeltaEncoder, DeltaDecoder, BooleanEncoder, BooleanDecoder } from "./encoding.mjs"
import { encodeChange, decodeChange, decodeDocument, decodeChanges } from "./columnar.mjs"
import { BloomFilter, encodeSyncMessage, encodeSyncState } from "./sync.mjs"
import fs from "fs"

const hex = b => Buffer.from(b).toString("hex")
let seed = 0xA070
function rnd() { // xorshift32
  seed ^= seed << 13; seed >>>= 0; seed ^= seed >>> 17; seed ^= seed << 5; seed >>>= 0
  return seed / 4294967296
}
const ri = n => Math.floor(rnd() * n)

const mode = process.argv[2]

if (mode === "codecs") {
  const out = []
  // LEB128
  const lebs = []
  for (const v of [0, 1, 63, 64, 127, 128, 255, 300, 16383, 16384, 2 ** 31, 2 ** 32, 2 ** 40 + 5, 2 ** 53 - 1]) {
    const e = new Encoder(); e.appendUint53(v); lebs.push({ kind: "uleb", value: v, hex: hex(e.buffer) })
  }
  for (const v of [0, 1, -1, 63, 64, -64, -65, 127, 128, -128, -129, 8191, 8192, -8192, -8193, 2 ** 31, -(2 ** 31), 2 ** 40 + 7, -(2 ** 40) - 7, 2 ** 53 - 1, -(2 ** 53) + 1]) {
    const e = new Encoder(); e.appendInt53(v); lebs.push({ kind: "sleb", value: v, hex: hex(e.buffer) })
  }
  out.push({ leb: lebs })
  const cases = []
  function genSeq(n, pNull, pRepeat, gen) {
    const vals = []
    let last = gen()
    for (let i = 0; i < n; i++) {
      const r = rnd()
      if (r < pNull) vals.push(null)
      else if (r < pNull + pRepeat) vals.push(last)
      else { last = gen(); vals.push(last) }
    }
    return vals
  }
  const shapes = [[0, 0, 0], [1, 0, 0], [1, 1, 0], [5, 1, 0], [5, 0, 1], [7, 0, 0], [40, 0.2, 0.5], [40, 0.5, 0.3], [60, 0.05, 0.1], [60, 0.0, 0.9], [100, 0.3, 0.0], [200, 0.1, 0.6]]
  for (let rep = 0; rep < 3; rep++) for (const [n, pn, pr] of shapes) {
    let vals = genSeq(n, pn, pr, () => ri(4) === 0 ? ri(2 ** 31) * 1024 + ri(1024) : ri(200))
    let e = new RLEEncoder("uint"); for (const v of vals) e.appendValue(v); cases.push({ kind: "rle_uint", values: vals, hex: hex(e.buffer) })
    vals = genSeq(n, pn, pr, () => (ri(4) === 0 ? ri(2 ** 31) * 1024 : ri(200)) * (ri(2) ? 1 : -1))
    e = new RLEEncoder("int"); for (const v of vals) e.appendValue(v); cases.push({ kind: "rle_int", values: vals, hex: hex(e.buffer) })
    vals = genSeq(n, pn, pr, () => ["", "a", "key", "héllo", "日本語", "k" + ri(50), "x".repeat(ri(40))][ri(7)])
    e = new RLEEncoder("utf8"); for (const v of vals) e.appendValue(v); cases.push({ kind: "rle_utf8", values: vals, hex: hex(e.buffer) })
    // deltas: mostly increasing counters
    let cur = 0
    vals = genSeq(n, pn, 0, () => { cur += ri(10) === 0 ? ri(100000) - 50000 : 1; return Math.max(cur, 0) })
    e = new DeltaEncoder(); for (const v of vals) e.appendValue(v); cases.push({ kind: "delta", values: vals, hex: hex(e.buffer) })
    const bools = []; let b = ri(2) === 1
    for (let i = 0; i < n; i++) { if (rnd() > pr) b = !b; bools.push(b) }
    e = new BooleanEncoder(); for (const v of bools) e.appendValue(v); cases.push({ kind: "boolean", values: bools, hex: hex(e.buffer) })
  }
  // decode direction sanity (JS decoder reading its own bytes), so the fixture is self-consistent
  for (const c of cases) {
    const buf = Buffer.from(c.hex, "hex")
    let d, got = []
    if (c.kind === "rle_uint") d = new RLEDecoder("uint", buf)
    else if (c.kind === "rle_int") d = new RLEDecoder("int", buf)
    else if (c.kind === "rle_utf8") d = new RLEDecoder("utf8", buf)
    else if (c.kind === "delta") d = new DeltaDecoder(buf)
    else d = new BooleanDecoder(buf)
    for (let i = 0; i < c.values.length; i++) got.push(d.readValue())
    if (JSON.stringify(got) !== JSON.stringify(c.values)) throw new Error("self check failed " + c.kind)
  }
  out.push({ columns: cases })
  fs.writeFileSync(process.argv[3], JSON.stringify(out))
}

// -------------------------------------------------------------------------------------------
// change scenarios: hand-authored op lists restating reference tests with FIXED actors
// (rust/automerge/tests/test.rs line numbers in `ref`), encoded by the reference's encodeChange
// -------------------------------------------------------------------------------------------
const A1 = "01" + "aa".repeat(15), A2 = "02" + "bb".repeat(15), A3 = "03" + "cc".repeat(15)
function mk(actor, seq, startOp, deps, ops, extra) {
  const c = Object.assign({ actor, seq, startOp, time: 0, message: "", deps, ops }, extra || {})
  const bin = encodeChange(c)
  const dec = decodeChange(bin)
  return { json: c, hex: hex(bin), hash: dec.hash }
}
const put = (obj, key, value, pred, datatype) => Object.assign({ action: "set", obj, key, insert: false, value, pred: pred || [] }, datatype ? { datatype } : {})
const del = (obj, key, pred) => ({ action: "del", obj, key, insert: false, pred })
const make = (kind, obj, key, pred) => ({ action: kind, obj, key, insert: false, pred: pred || [] })
const ins = (obj, elemId, value, datatype) => Object.assign({ action: "set", obj, elemId, insert: true, value, pred: [] }, datatype ? { datatype } : {})
const lset = (obj, elemId, value, pred) => ({ action: "set", obj, elemId, insert: false, value, pred })
const ldel = (obj, elemId, pred) => ({ action: "del", obj, elemId, insert: false, pred })
const inc = (obj, key, value, pred) => ({ action: "inc", obj, key, insert: false, value, pred })

function scenarios(lo, hi) { // lo < hi bytewise
  const S = []
  { // test.rs:101 merge_concurrent_map_prop_updates
    const c1 = mk(lo, 1, 1, [], [put("_root", "foo", "bar")])
    const c2 = mk(hi, 1, 1, [], [put("_root", "hello", "world")])
    S.push({ name: "merge_concurrent_map_prop_updates", ref: "tests/test.rs:101", changes: [c1, c2],
      expect: { foo: ["bar"], hello: ["world"] } })
  }
  { // test.rs:130 add_concurrent_increments_of_same_property
    const c1 = mk(lo, 1, 1, [], [put("_root", "counter", 0, [], "counter")])
    const c2 = mk(lo, 2, 2, [c1.hash], [inc("_root", "counter", 1, [`1@${lo}`])])
    const c3 = mk(hi, 1, 2, [c1.hash], [inc("_root", "counter", 2, [`1@${lo}`])])
    S.push({ name: "add_concurrent_increments_of_same_property", ref: "tests/test.rs:130", changes: [c1, c2, c3],
      expect: { counter: [{ counter: 3 }] } })
  }
  { // test.rs:178 concurrent_updates_of_same_field
    const c1 = mk(lo, 1, 1, [], [put("_root", "field", "one")])
    const c2 = mk(hi, 1, 1, [], [put("_root", "field", "two")])
    S.push({ name: "concurrent_updates_of_same_field", ref: "tests/test.rs:178", changes: [c1, c2],
      expect: { field: ["one", "two"] } })
  }
  { // test.rs:198 concurrent_updates_of_same_list_element
    const L = `1@${lo}`
    const c1 = mk(lo, 1, 1, [], [make("makeList", "_root", "birds"), ins(L, "_head", "finch")])
    const c2 = mk(lo, 2, 3, [c1.hash], [lset(L, `2@${lo}`, "greenfinch", [`2@${lo}`])])
    const c3 = mk(hi, 1, 3, [c1.hash], [lset(L, `2@${lo}`, "goldfinch", [`2@${lo}`])])
    S.push({ name: "concurrent_updates_of_same_list_element", ref: "tests/test.rs:198", changes: [c1, c2, c3],
      expect: { birds: [{ list: [["greenfinch", "goldfinch"]] }] } })
  }
  { // test.rs:385 concurrent_insertions_at_same_list_position
    const L = `1@${lo}`
    const c1 = mk(lo, 1, 1, [], [make("makeList", "_root", "birds"), ins(L, "_head", "parakeet")])
    const c2 = mk(lo, 2, 3, [c1.hash], [ins(L, `2@${lo}`, "starling")])
    const c3 = mk(hi, 1, 3, [c1.hash], [ins(L, `2@${lo}`, "chaffinch")])
    S.push({ name: "concurrent_insertions_at_same_list_position", ref: "tests/test.rs:385", changes: [c1, c2, c3],
      expect: { birds: [{ list: [["parakeet"], ["chaffinch"], ["starling"]] }] } })
  }
  { // test.rs:351 concurrent_insertions_at_different_list_positions
    const L = `1@${lo}`
    const c1 = mk(lo, 1, 1, [], [make("makeList", "_root", "list"), ins(L, "_head", "one"), ins(L, `2@${lo}`, "three")])
    const c2 = mk(lo, 2, 4, [c1.hash], [ins(L, `2@${lo}`, "two")])
    const c3 = mk(hi, 1, 4, [c1.hash], [ins(L, `3@${lo}`, "four")])
    S.push({ name: "concurrent_insertions_at_different_list_positions", ref: "tests/test.rs:351", changes: [c1, c2, c3],
      expect: { list: [{ list: [["one"], ["two"], ["three"], ["four"]] }] } })
  }
  { // test.rs:422 concurrent_assignment_and_deletion_of_a_map_entry
    const c1 = mk(lo, 1, 1, [], [put("_root", "bestBird", "robin")])
    const c2 = mk(lo, 2, 2, [c1.hash], [del("_root", "bestBird", [`1@${lo}`])])
    const c3 = mk(hi, 1, 2, [c1.hash], [put("_root", "bestBird", "magpie", [`1@${lo}`])])
    S.push({ name: "concurrent_assignment_and_deletion_of_a_map_entry", ref: "tests/test.rs:422", changes: [c1, c2, c3],
      expect: { bestBird: ["magpie"] } })
  }
  { // test.rs:443 concurrent_assignment_and_deletion_of_list_entry
    const L = `1@${lo}`
    const c1 = mk(lo, 1, 1, [], [make("makeList", "_root", "birds"), ins(L, "_head", "blackbird"), ins(L, `2@${lo}`, "thrush"), ins(L, `3@${lo}`, "goldfinch")])
    const c2 = mk(lo, 2, 5, [c1.hash], [lset(L, `3@${lo}`, "starling", [`3@${lo}`])])
    const c3 = mk(hi, 1, 5, [c1.hash], [ldel(L, `3@${lo}`, [`3@${lo}`])])
    S.push({ name: "concurrent_assignment_and_deletion_of_list_entry", ref: "tests/test.rs:443", changes: [c1, c2, c3],
      expect: { birds: [{ list: [["blackbird"], ["starling"], ["goldfinch"]] }] } })
  }
  { // test.rs:492 insertion_after_a_deleted_list_element
    const L = `1@${lo}`
    const c1 = mk(lo, 1, 1, [], [make("makeList", "_root", "birds"), ins(L, "_head", "blackbird"), ins(L, `2@${lo}`, "thrush"), ins(L, `3@${lo}`, "goldfinch")])
    const c2 = mk(lo, 2, 5, [c1.hash], [ldel(L, `3@${lo}`, [`3@${lo}`]), ldel(L, `4@${lo}`, [`4@${lo}`])])
    const c3 = mk(hi, 1, 5, [c1.hash], [ins(L, `3@${lo}`, "starling")])
    S.push({ name: "insertion_after_a_deleted_list_element", ref: "tests/test.rs:492", changes: [c1, c2, c3],
      expect: { birds: [{ list: [["blackbird"], ["starling"]] }] } })
  }
  { // test.rs:535 concurrent_deletion_of_same_list_element
    const L = `1@${lo}`
    const c1 = mk(lo, 1, 1, [], [make("makeList", "_root", "birds"), ins(L, "_head", "albatross"), ins(L, `2@${lo}`, "buzzard"), ins(L, `3@${lo}`, "cormorant")])
    const c2 = mk(lo, 2, 5, [c1.hash], [ldel(L, `3@${lo}`, [`3@${lo}`])])
    const c3 = mk(hi, 1, 5, [c1.hash], [ldel(L, `3@${lo}`, [`3@${lo}`])])
    S.push({ name: "concurrent_deletion_of_same_list_element", ref: "tests/test.rs:535", changes: [c1, c2, c3],
      expect: { birds: [{ list: [["albatross"], ["cormorant"]] }] } })
  }
  { // test.rs:651 does_not_interleave_sequence_insertions_at_same_position (doc1 = lo, doc2 = hi)
    const L = `1@${lo}`
    const c1 = mk(lo, 1, 1, [], [make("makeList", "_root", "wisdom")])
    const w1 = ["to", "be", "is", "to", "do"], w2 = ["to", "do", "is", "to", "be"]
    const ops1 = w1.map((w, i) => ins(L, i === 0 ? "_head" : `${i + 1}@${lo}`, w))
    const ops2 = w2.map((w, i) => ins(L, i === 0 ? "_head" : `${i + 1}@${hi}`, w))
    const c2 = mk(lo, 2, 2, [c1.hash], ops1)
    const c3 = mk(hi, 1, 2, [c1.hash], ops2)
    S.push({ name: "does_not_interleave_sequence_insertions_at_same_position", ref: "tests/test.rs:651", changes: [c1, c2, c3],
      expect: { wisdom: [{ list: w2.concat(w1).map(w => [w]) }] } })
  }
  { // test.rs:711/736 insertion at head by the other actor after merge: "one" goes before "two" either way
    const L = `1@${lo}`
    const c1 = mk(lo, 1, 1, [], [make("makeList", "_root", "list"), ins(L, "_head", "two")])
    const c2 = mk(hi, 1, 3, [c1.hash], [ins(L, "_head", "one")])
    S.push({ name: "multiple_insertions_at_same_list_position_a", ref: "tests/test.rs:711", changes: [c1, c2],
      expect: { list: [{ list: [["one"], ["two"]] }] } })
    const M = `1@${hi}`
    const d1 = mk(hi, 1, 1, [], [make("makeList", "_root", "list"), ins(M, "_head", "two")])
    const d2 = mk(lo, 1, 3, [d1.hash], [ins(M, "_head", "one")])
    S.push({ name: "multiple_insertions_at_same_list_position_b", ref: "tests/test.rs:736", changes: [d1, d2],
      expect: { list: [{ list: [["one"], ["two"]] }] } })
  }
  { // test.rs:761 insertion_consistent_with_causality
    const L = `1@${lo}`
    const c1 = mk(lo, 1, 1, [], [make("makeList", "_root", "list"), ins(L, "_head", "four")])
    const c2 = mk(hi, 1, 3, [c1.hash], [ins(L, "_head", "three")])
    const c3 = mk(lo, 2, 4, [c2.hash], [ins(L, "_head", "two")])
    const c4 = mk(hi, 2, 5, [c3.hash], [ins(L, "_head", "one")])
    S.push({ name: "insertion_consistent_with_causality", ref: "tests/test.rs:761", changes: [c1, c2, c3, c4],
      expect: { list: [{ list: [["one"], ["two"], ["three"], ["four"]] }] } })
  }
  { // datatypes, nested objects, text, table, bytes-ish (legacy JS has no bytes datatype on this path)
    const M = `2@${lo}`, T = `6@${lo}`
    const c1 = mk(lo, 1, 1, [], [
      put("_root", "n", null), make("makeMap", "_root", "m"), put(M, "i", -42, [], "int"), put(M, "u", 42, [], "uint"),
      put(M, "f", 3.25, [], "float64"), make("makeText", "_root", "t"), ins(T, "_head", "h"), ins(T, `7@${lo}`, "i"),
      put("_root", "ts", 1700000000000, [], "timestamp"), put("_root", "b", true), put("_root", "c", 10, [], "counter"),
      make("makeTable", "_root", "tbl"), put("_root", "big", 2 ** 40 + 123, [], "uint"), put("_root", "neg", -(2 ** 40), [], "int"),
    ], { message: "first", time: 1234567 })
    const c2 = mk(hi, 1, 15, [c1.hash], [
      inc("_root", "c", 5, [`11@${lo}`]), ins(T, `8@${lo}`, "!"), put(M, "i", 7, [`3@${lo}`], "int"), del("_root", "b", [`10@${lo}`]),
      ldel(T, `7@${lo}`, [`7@${lo}`]),
    ], { message: "second", time: -5, extraBytes: new Uint8Array([1, 2, 3]) })
    S.push({ name: "datatypes_nested", ref: "(own scenario)", changes: [c1, c2], expect: null })
  }
  return S
}

if (mode === "changes") {
  const out = { kat71: null, orders: [] }
  out.orders.push({ lo: A1, hi: A2, scenarios: scenarios(A1, A2) })
  out.orders.push({ lo: A2, hi: A3, scenarios: scenarios(A2, A3) })
  // a randomised batch of map changes for round-trip pinning
  const rand = []
  const keys = []; for (let i = 0; i < 12; i++) keys.push("k" + String(i).padStart(3, "0"))
  for (let t = 0; t < 30; t++) {
    const ops = []
    const n = 1 + ri(25)
    const start = 1 + ri(1000)
    for (let i = 0; i < n; i++) {
      const r = ri(10)
      const k = keys[ri(keys.length)]
      const pred = ri(3) === 0 ? [] : [`${1 + ri(start)}@${[A1, A2, A3][ri(3)]}`]
      if (r < 4) ops.push(put("_root", k, ri(1000) - 500, pred, "int"))
      else if (r < 6) ops.push(put("_root", k, "s" + "x".repeat(ri(20)), pred))
      else if (r < 7) ops.push(put("_root", k, ri(2) === 0, pred))
      else if (r < 8 && pred.length) ops.push(del("_root", k, pred))
      else if (r < 9) ops.push(put(`${1 + ri(start)}@${A2}`, k, ri(2 ** 30) * 1000, pred, "uint"))
      else ops.push(put("_root", k, rnd() * 1000, pred, "float64"))
    }
    const deps = []
    const nd = ri(3); for (let i = 0; i < nd; i++) { let h = ""; for (let j = 0; j < 64; j++) h += "0123456789abcdef"[ri(16)]; deps.push(h) }
    deps.sort()
    rand.push(mk([A1, A2, A3][ri(3)], 1 + ri(50), start, deps, ops, { time: ri(2 ** 31), message: ri(2) ? "m" + t : "" }))
  }
  out.random = rand
  fs.writeFileSync(process.argv[3], JSON.stringify(out))
}

if (mode === "decode_docs") {
  // argv[3] = directory of *.automerge written by our oracle; argv[4] = output json
  const dir = process.argv[3]
  const out = {}
  for (const f of fs.readdirSync(dir).filter(f => f.endsWith(".automerge")).sort()) {
    const buf = fs.readFileSync(dir + "/" + f)
    const changes = decodeDocument(buf) // reconstructs changes, re-encodes them and verifies heads
    out[f] = changes.map(c => { const bin = encodeChange(c); return { hash: decodeChange(bin).hash, hex: hex(bin), actor: c.actor, seq: c.seq, startOp: c.startOp, nops: c.ops.length } })
  }
  fs.writeFileSync(process.argv[4], JSON.stringify(out))
}

if (mode === "encode_change") {
  // argv[3] = one change as JSON (shape of `mk` above); argv[4] = output {hash, hex}
  const c = JSON.parse(fs.readFileSync(process.argv[3], "utf8"))
  const bin = encodeChange(c)
  fs.writeFileSync(process.argv[4], JSON.stringify({ hash: decodeChange(bin).hash, hex: hex(bin) }))
}

if (mode === "decode_doc") {
  // argv[3] = a document written by our oracle; argv[4] = output: per re-derived change its hash and op count
  const changes = decodeDocument(fs.readFileSync(process.argv[3]))
  fs.writeFileSync(process.argv[4], JSON.stringify(changes.map(c => ({ hash: decodeChange(encodeChange(c)).hash, actor: c.actor, seq: c.seq, startOp: c.startOp, nops: c.ops.length }))))
}

if (mode === "sync") {
  // the sync protocol's wire forms, from the reference's legacy JS implementation (javascript/test/legacy/sync.js):
  // Bloom filters over given hashes with probe results, encoded messages, encoded states
  const rh = () => { let h = ""; for (let j = 0; j < 64; j++) h += "0123456789abcdef"[ri(16)]; return h }
  const hashes = n => { const out = []; for (let i = 0; i < n; i++) out.push(rh()); return out }
  const out = { blooms: [], messages: [], states: [] }
  for (const n of [0, 1, 2, 3, 8, 17, 64, 200, 1000]) {
    const members = hashes(n)
    const f = new BloomFilter(members)
    const probes = hashes(200).map(h => ({ hash: h, contains: f.containsHash(h) }))
    for (const h of members) if (!f.containsHash(h)) throw new Error("member not found")
    out.blooms.push({ members, hex: hex(f.bytes), probes })
  }
  for (let t = 0; t < 24; t++) {
    const have = []
    const nh = [0, 1, 1, 2, 3][ri(5)]
    for (let i = 0; i < nh; i++) have.push({ lastSync: hashes(ri(4)).sort(), bloom: ri(4) === 0 ? new Uint8Array(0) : new BloomFilter(hashes(1 + ri(40))).bytes })
    const changes = []
    const nc = ri(5)
    for (let i = 0; i < nc; i++) { const l = [0, 1, 127, 128, 300, 20000][ri(6)]; const c = new Uint8Array(l); for (let j = 0; j < l; j++) c[j] = ri(256); changes.push(c) }
    const msg = { heads: hashes(ri(4)).sort(), need: hashes(ri(4)).sort(), have, changes }
    out.messages.push({ heads: msg.heads, need: msg.need, have: have.map(h => ({ lastSync: h.lastSync, bloom: hex(h.bloom) })), changes: changes.map(hex), hex: hex(encodeSyncMessage(msg)) })
  }
  for (let t = 0; t < 6; t++) {
    const sharedHeads = hashes(t).sort()
    out.states.push({ sharedHeads, hex: hex(encodeSyncState({ sharedHeads })) })
  }
  fs.writeFileSync(process.argv[3], JSON.stringify(out))
}
