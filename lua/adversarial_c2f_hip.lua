--[[ adversarial_c2f_hip.lua -- `adversarial.train(trainData)` and `adversarial.approxParzen(ds, nsamples, nneighbors)` of
adversarial_c2f.lua:10-223, 305-344 re-hosted on the step-level entries of libfacegen_hip.so in TABLE mode
(FG.Gan(dnG, dnD, true, B): G{noise[B,S,S,1], cond} through nn.JoinTable, D{x, cond} through nn.CAddTable).  Same signature, same
globals (OPT, OPTSTATE, MODEL_G, MODEL_D, IMG_DIMENSIONS, NOISE_DIM, COND_DIM, EPOCH, CONFUSION, best_dist), same loop: stride
batchSize/2, thisBatchSize at the tail, skip < 4, D_iterations x D-step (real picks .diff / .coarse, then NEW random .coarse picks
for the fake half -- adversarial_c2f.lua:126-143) then G_iterations x G-step (thisBatchSize .coarse picks, :165-169), timing prints,
confusion print, checkpoint every saveFreq epochs under the c2f file name.

The closures fevalD / fevalG_on_D (adversarial_c2f.lua:40-113) do not exist here: forward, BCECriterion, backward, penalty (incl. the
G_L2-as-L1-multiplier quirk of :103, kept inside fg_step_G), clamp and optimizer are one C call per closure.
`noiseInputs:uniform(-1, 1)` (:136, :163) is drawn on the device by the library's Philox stream (FG.manualSeed), not by torch's host
generator -- the one RNG stream that differs from the reference (DESIGN 5: parity is per step on the SAME inputs; tests feed them).

With MODELS.create_G(dims, false) (no GPU: train_c2f.lua --gpu -1) the nets carry no device plan and train() / approxParzen()
hand over to the reference's own adversarial_c2f.lua unchanged.

NOT EXECUTED IN THIS REPOSITORY'S ENVIRONMENT (no Lua / Torch7 in the image); face_generator_amd/adversarial_c2f.py is the executed
mirror (tests/test_gpu_train_epoch.py::test_adversarial_c2f_train_epoch_and_parzen compares it with the oracle's restatement of the
loop) and tests/test_lua_binding.py pins this file's call sites, pick order and save protocol against it.
Use from train_c2f.lua:  ADVERSARIAL = require 'adversarial_c2f_hip'   (lua/patches/train_c2f.lua.patch). ]]
local FG = require 'facegen_hip'
local C = FG.C
local adversarial = {}

-- {JoinTable | CAddTable, Copy, inner, Copy} of models_c2f.lua:113-145, 237-278: the device plan hangs on the inner Sequential
local function inner_of(model)
    local m = model.modules and model.modules[3]
    if m and m.fg then return m end
    return nil
end
local function reference_impl() return require 'adversarial_c2f' end

-- the step object lives as long as the two nets do
local function gan()
    if not adversarial.gan then
        adversarial.gan = FG.Gan(inner_of(MODEL_G).fg, inner_of(MODEL_D).fg, true, OPT.batchSize)
    end
    return adversarial.gan
end

-- `n` random examples' field (`diff` / `coarse` / `fine`, CHW FloatTensors) as one host batch; math.random order as the reference
local function pick(data, n, field, c, h, w)
    local out = torch.FloatTensor(n, c, h, w)
    for i = 1, n do out[i] = data[math.random(data:size())][field] end
    return out
end

function adversarial.train(trainData)
    if not (inner_of(MODEL_G) and inner_of(MODEL_D)) then return reference_impl().train(trainData) end
    EPOCH = EPOCH or 1
    local N_epoch = OPT.N_epoch
    if N_epoch <= 0 then N_epoch = trainData:size() end
    local dataBatchSize = OPT.batchSize / 2
    local time = sys.clock()
    local g = gan()
    local c, h, w = IMG_DIMENSIONS[1], IMG_DIMENSIONS[2], IMG_DIMENSIONS[3]
    local cc = COND_DIM[1]
    -- one device slot of 8 counts per D closure of the epoch, read once after the loop (no host sync, no allocation inside it)
    local maxClosures = (math.floor((N_epoch - 1) / dataBatchSize) + 1) * OPT.D_iterations
    local slots, nslots = FG.DeviceTensor(8 * maxClosures), 0

    print(string.format("<trainer> Epoch #%d [batchSize = %d]", EPOCH, OPT.batchSize))
    for t = 1, N_epoch, dataBatchSize do
        local thisBatchSize = math.min(OPT.batchSize, N_epoch - t + 1)
        if thisBatchSize < 4 then
            print(string.format("[INFO] skipping batch at t=%d, because its size is less than 4", t))
            break
        end
        thisBatchSize = thisBatchSize - thisBatchSize % 2
        local half = thisBatchSize / 2

        for k = 1, OPT.D_iterations do
            -- (1.1) real data: ONE pick gives .diff and .coarse (adversarial_c2f.lua:128-133)
            local diff, condR = torch.FloatTensor(half, c, h, w), torch.FloatTensor(half, cc, h, w)
            for i = 1, half do
                local ex = trainData[math.random(trainData:size())]
                diff[i] = ex.diff; condR[i] = ex.coarse
            end
            -- (1.2) sampled data: new random conditionings (:137-142); G's forward on them runs inside fg_step_D (TRAIN mode)
            local condF = pick(trainData, half, 'coarse', cc, h, w)
            g:configure('D', OPT, OPTSTATE)
            g:stepD(thisBatchSize, FG.to_device_nhwc(diff), FG.to_device_nhwc(condR), FG.to_device_nhwc(condF), false)
            g:confusionInto(slots, nslots); nslots = nslots + 1
        end

        for k = 1, OPT.G_iterations do
            local cond = pick(trainData, thisBatchSize, 'coarse', cc, h, w)      -- :165-169
            g:configure('G', OPT, OPTSTATE)
            g:stepG(thisBatchSize, FG.to_device_nhwc(cond), false)
        end
        xlua.progress(t + thisBatchSize, N_epoch)
    end
    local counts = {0, 0, 0, 0}
    for s = 0, nslots - 1 do                                  -- CONFUSION:add(c, targets[i] + 1) of :74-78, deferred
        local conf = FG.readConfusion(slots, s)
        for i = 1, 4 do counts[i] = counts[i] + conf[i] end
    end
    g:finishPending()

    time = sys.clock() - time
    print(string.format("<trainer> time required for this epoch = %d s", time))
    print(string.format("<trainer> time to learn 1 sample = %f ms", 1000 * time / N_epoch))
    print("Confusion of D:")
    for pred = 1, 2 do for target = 1, 2 do CONFUSION.mat[pred][target] = counts[(pred - 1) * 2 + target] end end
    CONFUSION:updateValids()
    print(CONFUSION)
    local tV = CONFUSION.totalValid
    CONFUSION:zero()

    if EPOCH % OPT.saveFreq == 0 then                         -- adversarial_c2f.lua:206-217
        local filename = paths.concat(OPT.save, string.format('adversarial_c2f_%d_to_%d.net', OPT.coarseSize, OPT.fineSize))
        os.execute(string.format("mkdir -p %s", sys.dirname(filename)))
        if paths.filep(filename) then os.execute(string.format("mv %s %s.old", filename, filename)) end
        print(string.format("<trainer> saving network to %s", filename))
        adversarial.save(filename, {D = MODEL_D, G = MODEL_G, opt = OPT, epoch = EPOCH})
    end
    EPOCH = EPOCH + 1
    return tV
end

-- torch.save with the device parameters brought back to the host modules and nothing unserialisable on them (FG.detach)
function adversarial.save(filename, tab)
    local innerD, innerG = inner_of(MODEL_D), inner_of(MODEL_G)
    innerD.fg:download(false); innerG.fg:download(false)
    NN_UTILS.prepareNetworkForSave(MODEL_G)
    NN_UTILS.prepareNetworkForSave(MODEL_D)
    local savedD, savedG = FG.detach(innerD), FG.detach(innerG)
    local ok, err = pcall(torch.save, filename, tab)
    FG.reattach(innerD, savedD); FG.reattach(innerG, savedG)
    if not ok then error(err) end
end

-- Unnormalized parzen window type estimate (adversarial_c2f.lua:305-344): nearest generation of `nneighbors` to the ground truth
function adversarial.approxParzen(ds, nsamples, nneighbors)
    if not inner_of(MODEL_G) then return reference_impl().approxParzen(ds, nsamples, nneighbors) end
    best_dist = best_dist or 1e10
    print('<trainer> evaluating approximate parzen ')
    local dnG = inner_of(MODEL_G).fg
    local c, h, w = IMG_DIMENSIONS[1], IMG_DIMENSIONS[2], IMG_DIMENSIONS[3]
    local cc = COND_DIM[1]
    local npix = nneighbors * h * w
    local noise, joined = FG.DeviceTensor(npix), FG.DeviceTensor(npix * (1 + cc))
    local dist_dev, min_dev = FG.DeviceTensor(nneighbors), FG.DeviceTensor(nsamples)
    adversarial.parzen_offset = adversarial.parzen_offset or 0
    for n = 1, nsamples do
        xlua.progress(n, nsamples)
        local example = ds[math.random(ds:size())]
        local condInputs = torch.FloatTensor(nneighbors, cc, h, w)
        for i = 1, nneighbors do condInputs[i] = example.coarse end
        local cond = FG.to_device_nhwc(condInputs)
        FG.check(C.fg_rng_uniform(FG.ctx, FG.seed, adversarial.parzen_offset, noise.ptr, npix, -1, 1))   -- noiseInputs:uniform(-1, 1)
        adversarial.parzen_offset = adversarial.parzen_offset + math.ceil(npix / 4)
        FG.check(C.fg_concat_channels(FG.ctx, noise.ptr, cond.ptr, joined.ptr, npix, 1, cc))              -- nn.JoinTable(2, 2)
        local neighbors = dnG:forward(joined, nneighbors)                                                   -- mode as the reference leaves it
        local fine = FG.to_device_nhwc(example.fine:float():view(1, c, h, w))
        -- neighbors:add(condInputs); dist = min_i torch.dist(neighbors[i], fine)  (:321-329): one reduction kernel, NHWC throughout
        FG.check(C.fg_parzen_min_dist(FG.ctx, neighbors, cond.ptr, fine.ptr, nneighbors, c * h * w, dist_dev.ptr, min_dev.ptr + (n - 1)))
    end
    local distances = min_dev:float()
    print('average || x_' .. OPT.fineSize .. ' - G(x_' .. OPT.coarseSize .. ') || = ' .. distances:mean())

    if distances:mean() < best_dist then                      -- :334-342
        best_dist = distances:mean()
        local filename = paths.concat(OPT.save, string.format('adversarial_c2f_%d_to_%d.bestnet', OPT.coarseSize, OPT.fineSize))
        os.execute('mkdir -p ' .. sys.dirname(filename))
        if paths.filep(filename) then os.execute('mv ' .. filename .. ' ' .. filename .. '.old') end
        print('<trainer> saving network to ' .. filename)
        adversarial.save(filename, {D = MODEL_D, G = MODEL_G, opt = OPT})
    end
    return distances
end

return adversarial
