--[[ adversarial_hip.lua -- `adversarial.train(dataset, maxAccuracyD, accsInterval)` of adversarial.lua:30-335 re-hosted on
the step-level entries of libfacegen_hip.so (fg_step_D / fg_step_G through lua/facegen_hip.lua).  Same signature, same
globals (OPT, OPTSTATE, MODEL_G, MODEL_D, IMG_DIMENSIONS, EPOCH, CONFUSION, Y_GENERATOR / Y_NOT_GENERATOR), same loop:
stride batchSize/2, thisBatchSize at the tail, skip < 4, D_iterations x D-step then G_iterations x G-step, the
maxAccuracyD gate over the last accsInterval batches, timing prints, confusion print, checkpoint every saveFreq epochs.

The closures fevalD / fevalG_on_D (adversarial.lua:83-231) do not exist here: their bodies -- forward, BCECriterion,
backward, penalty, clamp, optimizer -- are one C call each and never leave the GPU.  Deliberate deviation (SURVEY C6): the
G-step does not form D's weight gradients (the reference computes and discards them).

NOT EXECUTED IN THIS REPOSITORY'S ENVIRONMENT (no Lua / Torch7 in the image); face_generator_amd/adversarial.py is the
executed mirror of this file and tests/test_gpu_train_epoch.py compares it with the oracle's restatement of the loop.
Use from train.lua:  ADVERSARIAL = require 'adversarial_hip'   (lua/patches/train.lua.patch). ]]
local FG = require 'facegen_hip'
local C = FG.C
local adversarial = {}
adversarial.accs = {}

function adversarial.mean(t)            -- adversarial.lua:15-27
    local sum, count = 0, 0
    for _, v in pairs(t) do
        if type(v) == 'number' then sum = sum + v; count = count + 1 end
    end
    return sum / count
end

-- the step object lives as long as the two nets do
local function gan()
    if not adversarial.gan then
        local dnG, dnD = MODEL_G:get(2).fg, MODEL_D:get(2).fg       -- {Copy, net, Copy} of NN_UTILS.activateCuda
        adversarial.gan = FG.Gan(dnG, dnD, false, OPT.batchSize)
    end
    return adversarial.gan
end

function adversarial.train(dataset, maxAccuracyD, accsInterval)
    EPOCH = EPOCH or 1
    local N_epoch = OPT.N_epoch
    if N_epoch <= 0 then N_epoch = dataset:size() end
    local dataBatchSize = OPT.batchSize / 2
    local time = sys.clock()
    local g = gan()
    local countTrainedD, countNotTrainedD = 0, 0
    local c, h, w = IMG_DIMENSIONS[1], IMG_DIMENSIONS[2], IMG_DIMENSIONS[3]
    local useGate = maxAccuracyD <= 1.0          -- D_maxAcc = 1.01 (train.lua:37) can never fire: the update stays inside the step
    local counts = {0, 0, 0, 0}
    -- no gate: one device slot of 8 counts per D closure of the epoch, allocated ONCE here (hipMalloc is a blocking runtime call:
    -- none inside the loop) and read once after it
    local maxClosures = (math.floor((N_epoch - 1) / dataBatchSize) + 1) * OPT.D_iterations
    local slots, nslots = (not useGate) and FG.DeviceTensor(8 * maxClosures) or nil, 0

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
            -- (1.1) real data: math.random picks (adversarial.lua:244-249); (1.2) the fakes are drawn inside fg_step_D
            local real = torch.FloatTensor(half, c, h, w)
            for i = 1, half do real[i] = dataset[math.random(dataset:size())] end
            g:configure('D', OPT, OPTSTATE)
            g:stepD(thisBatchSize, FG.to_device_nhwc(real), nil, nil, useGate)
            if useGate then
                -- CONFUSION:add(c, targets[i] + 1) of adversarial.lua:112-117 and the gate of :124-178: the counts of the GLOBAL
                -- batch (host read, 32 bytes), the accuracy history, then the update -- or not
                local _, conf = g:confusion()                 -- [pred * 2 + target]
                for i = 1, 4 do counts[i] = counts[i] + conf[i] end
                local tV = (conf[1] + conf[4]) / math.max(1, conf[1] + conf[2] + conf[3] + conf[4])
                adversarial.accs[#adversarial.accs + 1] = tV  -- adversarial.lua:156-159
                if #adversarial.accs > accsInterval then table.remove(adversarial.accs, 1) end
                local doTrainD = adversarial.mean(adversarial.accs) < maxAccuracyD
                if doTrainD then g:update('D') end            -- not updating IS interruptableAdam's false,false path
                if doTrainD then countTrainedD = countTrainedD + 1 else countNotTrainedD = countNotTrainedD + 1 end
            else
                -- no gate to decide: the counts stay on the device (no host sync inside the epoch) and are read after the loop,
                -- exactly as face_generator_amd/adversarial.py defers them
                g:confusionInto(slots, nslots); nslots = nslots + 1
                countTrainedD = countTrainedD + 1
            end
        end

        for k = 1, OPT.G_iterations do
            g:configure('G', OPT, OPTSTATE)
            g:stepG(thisBatchSize, nil, false)
        end
        xlua.progress(t + thisBatchSize, N_epoch)
    end
    for s = 0, nslots - 1 do                                  -- deferred: one host read per D closure, after the epoch's last launch
        local conf = FG.readConfusion(slots, s)
        for i = 1, 4 do counts[i] = counts[i] + conf[i] end
        adversarial.accs[#adversarial.accs + 1] = (conf[1] + conf[4]) / math.max(1, conf[1] + conf[2] + conf[3] + conf[4])
        if #adversarial.accs > accsInterval then table.remove(adversarial.accs, 1) end
    end
    g:finishPending()

    time = sys.clock() - time
    print(string.format("<trainer> time required for this epoch = %d s", time))
    print(string.format("<trainer> time to learn 1 sample = %f ms", 1000 * time / N_epoch))
    print(string.format("<trainer> trained D %d of %d times.", countTrainedD, countTrainedD + countNotTrainedD))
    print("Confusion of normal D:")
    for pred = 1, 2 do for target = 1, 2 do CONFUSION.mat[pred][target] = counts[(pred - 1) * 2 + target] end end
    CONFUSION:updateValids()
    print(CONFUSION)
    local tV = CONFUSION.totalValid
    CONFUSION:zero()

    if EPOCH % OPT.saveFreq == 0 then                         -- adversarial.lua:319-329
        local filename = paths.concat(OPT.save, 'adversarial.net')
        os.execute(string.format("mkdir -p %s", sys.dirname(filename)))
        if paths.filep(filename) then os.execute(string.format("mv %s %s.old", filename, filename)) end
        print(string.format("<trainer> saving network to %s", filename))
        local innerD, innerG = MODEL_D:get(2), MODEL_G:get(2)
        innerD.fg:download(false); innerG.fg:download(false)                     -- device parameters -> the host modules
        NN_UTILS.prepareNetworkForSave(MODEL_D)
        NN_UTILS.prepareNetworkForSave(MODEL_G)
        -- the device plan (FFI cdata) and the overriding closures are not serialisable: off for the save, back on after it
        local savedD, savedG = FG.detach(innerD), FG.detach(innerG)
        local ok, err = pcall(torch.save, filename, {D = MODEL_D, G = MODEL_G, opt = OPT, epoch = EPOCH})
        FG.reattach(innerD, savedD); FG.reattach(innerG, savedG)
        if not ok then error(err) end
    end
    EPOCH = EPOCH + 1
    return tV
end

return adversarial
