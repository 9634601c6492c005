"""The LuaJIT binding (lua/*.lua) cannot be executed here (no Lua in the image).  What can be pinned mechanically is:
  * its ffi.cdef block is GENERATED from include/facegen_hip.h and up to date (scripts/gen_lua_cdef.py --check);
  * every `C.fg_*(...)` call names an entry the header declares, with the declared number of arguments;
  * every `C.FG_*` constant is an enumerator of the header;
  * every optimizer path tells the net that its parameters moved (the round-1 shim left the packed weights stale);
  * the shipped patches apply to the reference's own scripts (checked where /root/reference exists)."""
import glob
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LUA = sorted(glob.glob(os.path.join(ROOT, "lua", "*.lua")))


def strip_lua_comments(src):
    src = re.sub(r"--\[\[.*?\]\]", "", src, flags=re.S)
    return re.sub(r"--[^\n]*", "", src)


def lua_code(path):
    """Lua source without comments and without the cdef block."""
    src = open(path).read()
    src = re.sub(r"ffi\.cdef\[\[.*?\]\]", "", src, flags=re.S)
    return strip_lua_comments(src)


def call_sites(code, prefix="C.fg_"):
    """[(name, nargs)] for every `C.fg_xxx(...)` call: arguments counted at parenthesis depth 0."""
    out = []
    for m in re.finditer(r"\bC\.(fg_\w+)\s*\(", code):
        i, depth, nargs, seen = m.end(), 1, 0, False
        while depth > 0:
            ch = code[i]
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
            elif ch == "," and depth == 1:
                nargs += 1
            if depth > 0 and not ch.isspace():
                seen = True
            i += 1
        out.append((m.group(1), nargs + 1 if seen else 0))
    return out


def test_cdef_block_is_generated_from_the_header():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gen_lua_cdef.py"), "--check"])
    assert r.returncode == 0, "lua/facegen_hip.lua: cdef block is stale -- run python scripts/gen_lua_cdef.py"
    from face_generator_amd import _lib
    cdef = re.search(r"ffi\.cdef\[\[(.*?)\]\]", open(os.path.join(ROOT, "lua", "facegen_hip.lua")).read(), re.S).group(1)
    declared = set(re.findall(r"\b(fg_\w+)\s*\(", cdef))
    assert declared == set(_lib.parse_header()), declared ^ set(_lib.parse_header())


def test_every_call_site_matches_a_declaration():
    from face_generator_amd import _lib
    decls = _lib.parse_header()
    assert len(LUA) >= 2
    n = 0
    for path in LUA:
        for name, nargs in call_sites(lua_code(path)):
            assert name in decls, "%s calls C.%s, which include/facegen_hip.h does not declare" % (os.path.basename(path), name)
            want = len(decls[name][1])
            assert nargs == want, "%s: C.%s called with %d arguments, declared with %d" % (os.path.basename(path), name, nargs, want)
            n += 1
    assert n >= 40       # the binding is not a stub
    # entries that must be bound for the two levels of SURVEY 8(b) and for data parallelism
    used = {name for path in LUA for name, _ in call_sites(lua_code(path))}
    for must in ("fg_net_create", "fg_net_forward", "fg_net_backward", "fg_net_params_changed", "fg_step_D", "fg_step_G",
                 "fg_gan_update", "fg_gan_set_optimizer", "fg_comm_create", "fg_allreduce_sum", "fg_gan_set_comm",
                 "fg_concat_channels", "fg_add"):
        assert must in used, "the Lua binding never calls %s" % must


def test_constants_are_header_enumerators():
    hdr = open(os.path.join(ROOT, "include", "facegen_hip.h")).read()
    enums = set(re.findall(r"\b(FG_[A-Z0-9_]+)\s*=", hdr))
    for path in LUA:
        for c in re.findall(r"\bC\.(FG_[A-Z0-9_]+)\b", lua_code(path)):
            assert c in enums, "%s uses C.%s, not an enumerator of the header" % (os.path.basename(path), c)
    code = lua_code(os.path.join(ROOT, "lua", "facegen_hip.lua"))
    for must in ("FG_MAXPOOL2", "FG_CONV", "FG_STEP_NO_UPDATE", "FG_GAN_CONFUSION", "FG_PAUSED_SYNC"):
        assert "C." + must in code


def test_optimizers_mark_the_packed_weights_stale():
    code = lua_code(os.path.join(ROOT, "lua", "facegen_hip.lua"))
    for fn in ("interruptableAdam", "interruptableSgd", "interruptableAdagrad"):
        body = code[code.index("function M." + fn):]
        body = body[:body.index("\nend") + 4]
        assert "paramsChanged" in body, "M.%s does not call fg_net_params_changed after the update" % fn
    assert "function DeviceNet:paramsChanged() check(C.fg_net_params_changed(self.h)) end" in code
    up = code[code.index("function DeviceNet:upload()"):]
    assert "fg_net_params_changed" in up[:up.index("\nend")]
    # stride-2 convolutions, the max-pool and both table front ends are mapped
    assert "m.dW == 2" in code and "'nn.SpatialMaxPooling'" in code and "'nn.JoinTable'" in code and "fg_add" in code


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree is only present in the build container")
def test_patches_apply_to_the_reference_scripts(tmp_path):
    patches = sorted(glob.glob(os.path.join(ROOT, "lua", "patches", "*.patch")))
    assert len(patches) >= 6 and {"train_c2f.lua.patch", "models_c2f.lua.patch"} <= {os.path.basename(p) for p in patches}
    for p in patches:
        r = subprocess.run(["patch", "-p1", "--dry-run", "-d", "/root/reference", "-i", p], capture_output=True, text=True)
        assert r.returncode == 0, "%s does not apply: %s" % (os.path.basename(p), r.stdout + r.stderr)
    # after patching, no CUDA rock is required on the training / sampling path
    import shutil
    RELS = ("train.lua", "sample.lua", "utils/nn_utils.lua", "layers/cudnnSpatialConvolutionUpsample.lua", "train_c2f.lua", "models_c2f.lua",
            "dataset_c2f.lua")
    for rel in RELS:
        os.makedirs(os.path.dirname(os.path.join(str(tmp_path), rel)), exist_ok=True)
        shutil.copy(os.path.join("/root/reference", rel), os.path.join(str(tmp_path), rel))
    for p in patches:
        subprocess.run(["patch", "-p1", "-s", "-d", str(tmp_path), "-i", p], check=True)
    for rel in RELS:
        code = strip_lua_comments(open(os.path.join(str(tmp_path), rel)).read())
        assert not re.search(r"require\s+'(cutorch|cunn|cudnn)'", code), "%s still requires a CUDA rock" % rel
        assert "cutorch." not in code and ":cuda()" not in code, "%s still calls into cutorch" % rel
        assert "CudaTensor" not in code, "%s still names torch.CudaTensor" % rel
    # configs 4-5: every create_G_* / create_D_* variant of models_c2f.lua keeps its {first, Copy, inner, Copy}
    # shape and hands the inner Sequential to FG.attach with the dimensions its first layer takes; train_c2f.lua binds the library,
    # selects the re-hosted loop and re-attaches a reloaded checkpoint
    mc = strip_lua_comments(open(os.path.join(str(tmp_path), "models_c2f.lua")).read())
    assert mc.count("FG.attach(model_G:get(3), {dimensions[1]+1, dimensions[2], dimensions[3]}, OPT.batchSize)") == 4
    assert mc.count("FG.attach(model_D:get(3), {dimensions[1], dimensions[2], dimensions[3]}, OPT.batchSize)") == 3
    assert mc.count("nn.Copy('torch.FloatTensor', 'torch.FloatTensor')") == 14 and mc.count("nn.JoinTable(2, 2)") == 4
    # round 6: dataset._toResult (dataset_c2f.lua:49-61) hands image.scale down / up and the subtraction to the library; the `image`
    # package stays required for image.load only, and FG is bound (train_c2f.lua) before DATASET.loadImages runs
    dc = strip_lua_comments(open(os.path.join(str(tmp_path), "dataset_c2f.lua")).read())
    body = dc[dc.index("function dataset._toResult"):dc.index("local result = {}")]
    assert "FG.coarseDiff(fineImages, dataset.coarseScale)" in body and "image.scale" not in body
    assert dc.count("image.scale") == 2 and "image.load" in dc        # the JPEG loader (dataset_c2f.lua:111-215, out of scope) keeps its own
    assert "result.coarse = coarseImages" in dc and "result.diff = diffImages" in dc
    tc = strip_lua_comments(open(os.path.join(str(tmp_path), "train_c2f.lua")).read())
    assert tc.index("FG = require 'facegen_hip'") < tc.index("DATASET.loadImages(0, 500)")
    assert "ADVERSARIAL = require 'adversarial_c2f_hip'" in tc and "FG = require 'facegen_hip'" in tc
    assert tc.index("FG = require 'facegen_hip'") < tc.index("FG.setDevice(OPT.gpu + 1)") < tc.index("FG.manualSeed(OPT.seed)") \
        < tc.index("MODELS.create_D(IMG_DIMENSIONS, OPT.gpu ~= false)")
    assert "FG.attach(MODEL_G:get(3), {IMG_DIMENSIONS[1] + 1, IMG_DIMENSIONS[2], IMG_DIMENSIONS[3]}, OPT.batchSize)" in tc


def _fn_body(code, header):
    body = code[code.index(header):]
    return body[:body.index("\nend") + 4]


def test_device_side_updates_are_not_overwritten_by_a_module_level_forward():
    """adversarial_hip.lua moves only the DEVICE vectors; NN_UTILS.visualizeProgress then calls
    MODEL_G:forward / MODEL_D:forward (train.lua:204, nn_utils.lua:52, 96), whose override used to upload the stale host weights
    unconditionally.  Pinned: the override synchronises in the right direction, and every path that moves device parameters says so."""
    code = lua_code(os.path.join(ROOT, "lua", "facegen_hip.lua"))
    fwd = _fn_body(code, "function seq:updateOutput(input)")
    assert "dn:syncForModuleCall()" in fwd and "dn:upload()" not in fwd
    sync = _fn_body(code, "function DeviceNet:syncForModuleCall()")
    assert re.search(r"if self\.device_newer then self:download\(false\) else self:upload\(\) end", sync)
    assert "self.device_newer = false" in _fn_body(code, "function DeviceNet:upload()")
    assert "self.device_newer = false" in _fn_body(code, "function DeviceNet:download(want_grads)")
    stepD, stepG, upd = (_fn_body(code, "function Gan:stepD("), _fn_body(code, "function Gan:stepG("), _fn_body(code, "function Gan:update("))
    assert "self.G:markDeviceNewer()" in stepD and "if not hold then self.D:markDeviceNewer() end" in stepD
    assert "self.G:markDeviceNewer()" in stepG and "self.D:markDeviceNewer()" in stepG
    assert "self.D:markDeviceNewer()" in upd and "self.G:markDeviceNewer()" in upd
    for fn in ("interruptableAdam", "interruptableSgd", "interruptableAdagrad"):
        assert "net:markDeviceNewer()" in _fn_body(code, "function M." + fn)


def test_nothing_unserialisable_is_on_a_module_during_save_or_clone():
    """torch.save / Module:clone write every field of a module, closures with their upvalues included, and
    cannot write FFI cdata -- the device plan and the overrides come off around both."""
    code = lua_code(os.path.join(ROOT, "lua", "facegen_hip.lua"))
    attached = re.search(r"local ATTACHED = \{(.*?)\}", code).group(1)
    installed = set(re.findall(r"function seq:(\w+)\(", code)) | set(re.findall(r"\bseq\.(\w+)\s*=", code))
    assert installed and installed <= set(re.findall(r"'(\w+)'", attached)), "a field M.attach installs is not in ATTACHED: %s" % installed
    adv = lua_code(os.path.join(ROOT, "lua", "adversarial_hip.lua"))
    save = adv[adv.index("FG.detach(innerD)"):]
    assert save.index("FG.detach(innerG)") < save.index("torch.save") < save.index("FG.reattach(innerD, savedD)")
    patch = open(os.path.join(ROOT, "lua", "patches", "utils_nn_utils.lua.patch")).read()
    i = patch.index("FG.detach(inner)")
    assert i < patch.index("pcall(inner.clone, inner)") < patch.index("FG.reattach(inner, saved)")


def test_the_two_mirrors_of_adversarial_train_defer_the_confusion_counts_alike():
    """without a gate neither mirror reads the counts inside the epoch; with one both read per D closure."""
    adv = lua_code(os.path.join(ROOT, "lua", "adversarial_hip.lua"))
    # one per-epoch slot buffer instead of a hipMalloc per closure: nothing is allocated or read inside the loop
    assert "g:confusionInto(slots, nslots); nslots = nslots + 1" in adv and "FG.readConfusion(slots, s)" in adv
    loop = adv[adv.index("for t = 1, N_epoch, dataBatchSize do"):adv.index("for s = 0, nslots - 1 do")]
    assert "FG.DeviceTensor(" not in loop and "readConfusion" not in loop
    assert adv.index("FG.DeviceTensor(8 * maxClosures)") < adv.index("for t = 1, N_epoch, dataBatchSize do")
    py = open(os.path.join(ROOT, "face_generator_amd", "adversarial.py")).read()
    assert "pending.append(r[\"confusion\"].clone())" in py and "if not use_gate:" in py


def test_c2f_rehost_follows_adversarial_c2f_lua():
    """lua/adversarial_c2f_hip.lua against adversarial_c2f.lua:10-223, 305-344 and its executed Python mirror: table-mode step
    object, the reference's pick order (real .diff / .coarse from ONE pick, then new .coarse picks for the fake half, then
    thisBatchSize .coarse picks for the G closure), counts deferred without allocations, both checkpoint names, the parzen distance
    on the device, and the hand-over to the reference's own file when the nets carry no device plan."""
    adv = lua_code(os.path.join(ROOT, "lua", "adversarial_c2f_hip.lua"))
    assert "FG.Gan(inner_of(MODEL_G).fg, inner_of(MODEL_D).fg, true, OPT.batchSize)" in adv       # table_inputs = true
    assert "model.modules[3]" in adv                                                             # {first, Copy, inner, Copy}
    train = adv[adv.index("function adversarial.train(trainData)"):adv.index("function adversarial.save(")]
    assert "return reference_impl().train(trainData)" in train and "require 'adversarial_c2f'" in adv
    d_it = train[train.index("for k = 1, OPT.D_iterations do"):train.index("for k = 1, OPT.G_iterations do")]
    one_pick = d_it.index("local ex = trainData[math.random(trainData:size())]")
    assert one_pick < d_it.index("diff[i] = ex.diff; condR[i] = ex.coarse") < d_it.index("pick(trainData, half, 'coarse'") \
        < d_it.index("g:stepD(thisBatchSize, FG.to_device_nhwc(diff), FG.to_device_nhwc(condR), FG.to_device_nhwc(condF), false)")
    assert d_it.count("math.random") == 1 and "FG.DeviceTensor(" not in d_it                      # the second pick is pick()'s
    g_it = train[train.index("for k = 1, OPT.G_iterations do"):train.index("xlua.progress(")]
    assert g_it.index("pick(trainData, thisBatchSize, 'coarse'") < g_it.index("g:stepG(thisBatchSize, FG.to_device_nhwc(cond), false)")
    for must in ("thisBatchSize < 4", "thisBatchSize - thisBatchSize % 2", "g:configure('D', OPT, OPTSTATE)", "g:configure('G', OPT, OPTSTATE)",
                 "g:finishPending()", "CONFUSION:updateValids()", "EPOCH % OPT.saveFreq == 0", "'adversarial_c2f_%d_to_%d.net'",
                 "EPOCH = EPOCH + 1"):
        assert must in train, must
    save = adv[adv.index("function adversarial.save("):adv.index("function adversarial.approxParzen(")]
    assert save.index("FG.detach(innerD)") < save.index("pcall(torch.save, filename, tab)") < save.index("FG.reattach(innerD, savedD)")
    assert save.index("fg:download(false)") < save.index("FG.detach(innerD)")
    pz = adv[adv.index("function adversarial.approxParzen("):]
    for must in ("best_dist = best_dist or 1e10", "ds[math.random(ds:size())]", "C.fg_rng_uniform(", "C.fg_concat_channels(",
                 "dnG:forward(joined, nneighbors)", "C.fg_parzen_min_dist(", "min_dev.ptr + (n - 1)", "distances:mean() < best_dist",
                 "'adversarial_c2f_%d_to_%d.bestnet'", "return distances"):
        assert must in pz, must
    # the executed mirror makes the same calls in the same order
    py = open(os.path.join(ROOT, "face_generator_amd", "adversarial_c2f.py")).read()
    tr = py[py.index("def train(trainData):"):py.index("def approxParzen(")]
    assert tr.index("trainData[i].diff") < tr.index("trainData[i].coarse") < tr.index('pick(half, "coarse")') < tr.index("tr.step_D(diff, cond_r, nz, cond_f)")
    assert tr.index('pick(thisBatchSize, "coarse")') < tr.index("tr.step_G(nz, cond)")
    assert "fg_parzen_min_dist" in py[py.index("def approxParzen("):]
    binding = lua_code(os.path.join(ROOT, "lua", "facegen_hip.lua"))
    into = _fn_body(binding, "function Gan:confusionInto(t, slot)")
    assert "C.fg_d2d(ctx, t.ptr + 8 * slot" in into and "fg_malloc" not in into


def test_lua_files_are_block_balanced():
    """No Lua interpreter in the image: the least a syntax error could hide behind is checked mechanically -- every block opener
    (function / if / do / repeat) has its end / until, every bracket its partner (comments and string literals stripped)."""
    def strip(src):
        src = re.sub(r"--\[\[.*?\]\]", "", src, flags=re.S)
        src = re.sub(r"\[\[.*?\]\]", "''", src, flags=re.S)
        src = re.sub(r"--[^\n]*", "", src)
        src = re.sub(r"'(?:\\.|[^'\\\n])*'", "''", src)
        return re.sub(r'"(?:\\.|[^"\\\n])*"', '""', src)
    for path in LUA:
        src = strip(open(path).read())
        depth = 0
        for t in re.findall(r"\b(function|if|do|repeat|until|end)\b", src):
            depth += 1 if t in ("function", "if", "do", "repeat") else -1
            assert depth >= 0, "%s: an `end` without an opener" % os.path.basename(path)
        assert depth == 0, "%s: %d unclosed block(s)" % (os.path.basename(path), depth)
        for a, b in ("()", "{}", "[]"):
            assert src.count(a) == src.count(b), "%s: unbalanced %s%s" % (os.path.basename(path), a, b)


# ---------------------------------------------------------------------------------------------------------------------------
# no Lua in the image, so the shipped Lua is syntax-checked by a Lua 5.1 parser written for the purpose
# (tests/lua_parser.py: the whole grammar, not a block counter) and its free names are resolved against what the host script defines.
# ---------------------------------------------------------------------------------------------------------------------------
from lua_parser import parse as lua_parse, LuaSyntaxError, tokenize as lua_tokenize      # noqa: E402

LUA_STDLIB = {"assert", "collectgarbage", "dofile", "error", "getfenv", "getmetatable", "ipairs", "load", "loadfile", "loadstring",
              "module", "next", "pairs", "pcall", "print", "rawequal", "rawget", "rawset", "require", "select", "setfenv",
              "setmetatable", "tonumber", "tostring", "type", "unpack", "xpcall", "_G", "_VERSION", "coroutine", "debug", "io", "math",
              "os", "package", "string", "table", "bit", "jit", "arg"}


def test_the_parser_knows_lua():
    """The checker itself: accepts the constructs of the 5.1 manual, rejects the classic slips, resolves scopes."""
    ok = r'''
        local a, b = 1, 0x1F; local s = 'x\'y' .. "z" .. [[long
        string]] .. [==[ with ]] inside ]==]
        --[[ long
        comment ]] --[==[ another ]==]
        local function f(x, ...) local t = {...}; return x and #t or -x ^ 2 ^ 3, select('#', ...) end
        function M.sub.name:method(p) self.v = p; return self end
        for i = 1, 10, 2 do if i % 2 == 0 then break elseif i > 5 then g = i else h = {i, [i] = 2, k = 3; 4,} end end
        for k, v in pairs(t) do repeat local z = k until z ~= nil end
        while not done do done = f(a)(b){c}'d':m(1, 2):n "s" [1].x end
        do local x <const_is_not_5_1 = 1 end
    '''
    with pytest.raises(LuaSyntaxError):
        lua_parse(ok)                                # `<attrib>` is 5.4, not 5.1: rejected
    c = lua_parse(ok.replace(" <const_is_not_5_1", ""))
    assert {"M", "pairs", "t", "done", "select"} <= set(c.globals_read) | set(c.globals_written)
    assert "a" not in c.globals_read and "f" not in c.globals_read                     # locals (incl. `local function`) are not free
    assert {"g", "h", "done"} <= set(c.globals_written) and "z" not in c.globals_read and "self" not in c.globals_read
    assert ("M.sub.name:method", 1, False) in [(q, n, v) for (_, q, n, v) in c.functions]
    for bad in ("if x then y = 1", "if x y = 1 end", "f(1,,2)", "x = = 1", "local function end", "for i = 1 do end", "x = {1 2}",
                "return 1 x = 2", "break", "f() = 1", "a.b:c = 1", "x = 1 +", "function f(a,) end", "local t = {[1] 2}",
                "repeat x = 1 end", "y = 'unfinished", "z = [[unfinished", "x = 3e", "local function f() return ... end",
                "goto done", "x = a ? b : c", "end"):
        with pytest.raises(LuaSyntaxError):
            lua_parse(bad)
    assert lua_parse("local x = 1ULL + 0x10LL + 2i").n_statements == 1                    # LuaJIT number suffixes (the FFI binding uses them)


def test_shipped_lua_parses():
    assert len(LUA) >= 3
    for path in LUA:
        c = lua_parse(open(path).read())
        assert c.n_statements > 50 and c.functions, os.path.basename(path)


def _mutations(src):
    """Three deliberate slips per file: a dropped `end`, a dropped `then`, a doubled comma in an argument list."""
    toks = lua_tokenize(src)
    lines = src.split("\n")

    def drop(word, which):
        hits = [t for t in toks if t.kind == "keyword" and t.val == word]
        t = hits[which % len(hits)]
        ln = lines[t.line - 1]
        i = [m.start() for m in re.finditer(r"\b%s\b" % word, ln)]
        # the token stream has no columns: take the line's last occurrence that is not inside a comment
        code = ln.split("--")[0]
        i = [k for k in i if k < len(code)]
        assert i, (word, t.line, ln)
        out = list(lines)
        out[t.line - 1] = ln[:i[-1]] + " " * len(word) + ln[i[-1] + len(word):]
        return "\n".join(out)
    yield "a dropped `end`", drop("end", -1)
    yield "a dropped `end` mid-file", drop("end", len([t for t in toks if t.val == "end"]) // 2)
    yield "a dropped `then`", drop("then", 1)
    blank = lambda mm: re.sub(r"[^\n]", " ", mm.group(0))                  # comments blanked in place: offsets stay valid
    code = re.sub(r"--[^\n]*", blank, re.sub(r"--\[(=*)\[.*?\]\1\]", blank, src, flags=re.S))
    code = re.sub(r"\[(=*)\[.*?\]\1\]", blank, code, flags=re.S)             # long strings too (the cdef block is C, not Lua)
    m = re.search(r"\w\(([^()\n'\"]*?), ", code)
    yield "a doubled comma", src[:m.end() - 1] + ", " + src[m.end() - 1:]


def test_a_broken_end_then_or_comma_is_caught_in_every_file():
    for path in LUA:
        src = open(path).read()
        lua_parse(src)
        for what, bad in _mutations(src):
            assert bad != src
            with pytest.raises(LuaSyntaxError):
                lua_parse(bad)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree is only present in the build container")
def test_patched_reference_scripts_parse_and_define_what_the_rehosted_files_read(tmp_path):
    """Every reference script, untouched and after lua/patches/*.patch, is valid Lua 5.1; and every free name the re-hosted loops
    read is a Lua / LuaJIT library name, a global the (patched) host script assigns before it calls the loop, or a name the
    reference's own version of that file reads as well (train.lua:71-94 sets OPT, MODEL_G, NN_UTILS, CONFUSION, ...)."""
    import shutil
    ref = "/root/reference"
    scripts = sorted(glob.glob(os.path.join(ref, "*.lua")) + glob.glob(os.path.join(ref, "*", "*.lua")))
    assert len(scripts) >= 15
    for s in scripts:
        lua_parse(open(s).read())                                                   # the checker accepts all of upstream's Lua
        rel = os.path.relpath(s, ref)
        os.makedirs(os.path.dirname(os.path.join(str(tmp_path), rel)) or str(tmp_path), exist_ok=True)
        shutil.copy(s, os.path.join(str(tmp_path), rel))
    for p in sorted(glob.glob(os.path.join(ROOT, "lua", "patches", "*.patch"))):
        subprocess.run(["patch", "-p1", "-s", "-d", str(tmp_path), "-i", p], check=True)
    parsed = {}
    for s in scripts:
        rel = os.path.relpath(s, ref)
        parsed[rel] = lua_parse(open(os.path.join(str(tmp_path), rel)).read())      # ... and all of it after the patches
    mine = {os.path.basename(p): lua_parse(open(p).read()) for p in LUA}
    binding = mine["facegen_hip.lua"]
    extra = set(binding.globals_read) - LUA_STDLIB - {"torch", "IMG_DIMENSIONS"}
    assert not extra, "lua/facegen_hip.lua reads undefined globals: %s" % sorted(extra)
    for host, loop, upstream in (("train.lua", "adversarial_hip.lua", "adversarial.lua"),
                                 ("train_c2f.lua", "adversarial_c2f_hip.lua", "adversarial_c2f.lua")):
        defined = set(parsed[host].globals_written) | set(mine[loop].globals_written)
        theirs = set(lua_parse(open(os.path.join(ref, upstream)).read()).globals_read)
        unknown = set(mine[loop].globals_read) - LUA_STDLIB - defined - theirs
        assert not unknown, "%s reads %s, which %s never assigns and %s never reads" % (loop, sorted(unknown), host, upstream)
        # the patched host still assigns everything the loop reads from it, and binds the library before building the nets
        for name in ("OPT", "MODEL_G", "MODEL_D", "NN_UTILS", "CONFUSION", "OPTSTATE", "EPOCH", "IMG_DIMENSIONS", "ADVERSARIAL"):
            assert name in parsed[host].globals_written, "%s (patched) no longer assigns %s" % (host, name)
        assert "FG" in parsed[host].globals_written, "%s (patched) does not bind the library (FG = require 'facegen_hip')" % host
