'use strict';
/*
 * HipRenderer: drop-in for the reference's WebGPURenderer
 * (src/webgpu-renderer.ts:20-322) on an AMD MI355X through librt_hip.so.
 * Same construction contract (static async Create resolving to a renderer or
 * null, never rejecting), same controls, same frame sequencing (frameIndex is
 * incremented before the uniforms are written), same counter semantics.
 * `canvas` only needs width and height; an optional `pixels` typed array on it
 * receives the presented image.
 *
 * Added for headless use and parity checks (the reference never reads images
 * back): renderFrames(n), sync(), readAccum(), readIds(), readFrameF16(),
 * getCounters(), destroy().
 */
const path = require('path');
const Common = require('./common');
const { RethrownError } = Common;
const { Accumulator } = require('./accumulator');
const { FPSCounter } = require('./fps-counter');
const Schemas = require('./gpu-schemas');

function loadNative() {
    return require(path.join(__dirname, 'rt_hip_napi.node'));
}

class HipRenderer {
    constructor(canvas) {
        // private in the reference (src/webgpu-renderer.ts:42-46)
        this.canvas = canvas;
        this.resolution = { width: canvas.width, height: canvas.height };
        this.native = null;
        this.handle = null;
        this.accumulator = null;
        this.currentScene = null;
        this.frameIndex = 0;
        this.fpsCounter = new FPSCounter();
        this.samplesPerPixel = 1;
        this.debugEnabled = false;
        this._enabled = true;
        this._loopTimer = null;
    }

    // src/webgpu-renderer.ts:48-103.  options: { device, maxBounces, stripeRows, stripeIndex, stripeCount,
    // presentEvery, applyModelTransforms, sphereLights, debugView }: sphereLights samples emissive spheres too (the shader's
    // unused sample_light_spheres; off by default, changes the image); debugView makes setDebug(true) draw every
    // triangle's normal marker (the shader's unused ray_triangle_normal_intersect; off by default: the reference's
    // setDebug changes nothing on screen); applyModelTransforms applies a model's scale / rotation / position, which the
    // reference reads and ignores (off by default); presentEvery = k copies the image into canvas.pixels on every k-th frame only (default 1,
    // every frame as the reference's canvas pass does; like that pass it does not wait for the GPU: canvas.pixels shows the newest
    // image that has arrived, a launch or two behind the frames submitted; presentNow() waits for the current one)
    static async Create(canvas, scenePath, options) {
        const renderer = new HipRenderer(canvas);
        const opt = options || {};
        try {
            try {
                renderer.native = loadNative();
                if (renderer.native.deviceCount() < 1) throw new Error('No HIP device found');
                renderer.handle = renderer.native.create(
                    opt.device || 0, renderer.resolution.width, renderer.resolution.height,
                    opt.stripeRows || 8, opt.stripeIndex || 0, opt.stripeCount || 1);
                if (opt.maxBounces !== undefined) renderer.native.setMaxBounces(renderer.handle, opt.maxBounces);
                renderer.presentEvery = Math.max(1, opt.presentEvery | 0);
                if (opt.sphereLights) renderer.native.setSphereLights(renderer.handle, 1);
                if (opt.debugView) renderer.native.setDebugView(renderer.handle, 1);
                renderer.stripe = { rows: opt.stripeRows || 8, index: opt.stripeIndex || 0, count: opt.stripeCount || 1 };
            } catch (error) {
                throw new RethrownError('Failed to create pipelines', error);
            }
            try {
                renderer.currentScene = await Common.loadScene(scenePath, { applyModelTransforms: !!opt.applyModelTransforms });
            } catch (error) {
                throw new RethrownError("Failed to load '" + scenePath + "'", error);
            }
            try {
                await renderer.createBuffersAndBindGroups();
            } catch (error) {
                throw new RethrownError('Failed to create scene buffer', error);
            }
        } catch (error) {
            Common.showError(error.message);
            return null;
        }
        return renderer;
    }

    // src/webgpu-renderer.ts:141-212
    async createBuffersAndBindGroups() {
        if (!this.handle) throw new Error('Device not initialized');
        if (!this.currentScene) throw new Error('Scene not loaded');
        const scene = this.currentScene;
        this.native.setScene(this.handle, Schemas.packSpheres(scene.spheres), Schemas.packTriangles(scene.triangles));
        this.writeUniforms();
        this.accumulator = await Accumulator.create({ native: this.native, handle: this.handle }, this.resolution);
    }

    // src/webgpu-renderer.ts:214-230
    writeUniforms() {
        if (!this.handle || !this.currentScene) return;
        this.native.setUniforms(this.handle, Schemas.packUniforms(
            this.currentScene.camera, this.frameIndex, this.resolution.width, this.resolution.height,
            this.samplesPerPixel, this.debugEnabled));
    }

    // src/webgpu-renderer.ts:232-235
    setSamplesPerPixel(samples) {
        this.samplesPerPixel = Math.max(1, Math.min(16, samples));
        this.writeUniforms();
    }

    // src/webgpu-renderer.ts:237-268: no-op without resources; fire and forget
    render() {
        if (!this.handle || !this.accumulator) return;
        this.frameIndex++;
        this.writeUniforms();
        this.native.renderFrame(this.handle);
        const present = this.canvas.pixels && this.frameIndex % (this.presentEvery || 1) === 0;
        this.accumulator.render(null, present ? this.canvas.pixels : null);
    }

    // n calls of render() as one native submission
    renderFrames(n) {
        if (!this.handle || !this.accumulator || n <= 0) return;
        const first = this.frameIndex + 1;
        this.frameIndex += n;
        this.native.renderFrames(this.handle, first, n);
        this.writeUniforms();
        const every = this.presentEvery || 1;
        const present = this.canvas.pixels && Math.floor(this.frameIndex / every) > Math.floor((first - 1) / every);
        this.accumulator.render(null, present ? this.canvas.pixels : null);
    }

    // src/webgpu-renderer.ts:270-277 with setImmediate in place of requestAnimationFrame
    startRenderLoop() {
        const loop = () => {
            this.fpsCounter.updateFPS();
            // no fence, as in the reference (src/webgpu-renderer.ts:267): the native render call
            // itself waits when two frames are already in flight
            if (this._enabled) this.render();
            this._loopTimer = setImmediate(loop);
        };
        this._loopTimer = setImmediate(loop);
    }

    stopRenderLoop() {
        if (this._loopTimer) clearImmediate(this._loopTimer);
        this._loopTimer = null;
    }

    setDebug(checked) {
        this.debugEnabled = checked;
        this.writeUniforms();
    }

    // src/webgpu-renderer.ts:284-287: frameIndex only; the planes keep their contents
    resetAccumulation() {
        this.frameIndex = 0;
        this.writeUniforms();
    }

    getFPS() {
        return this.fpsCounter.getFPS();
    }

    getEnabled() {
        return this._enabled;
    }

    setEnabled(value) {
        this._enabled = value;
    }

    // src/webgpu-renderer.ts:301-321: u32 counter, zeroed by the read
    async getTriangleTestsSinceLastCheck() {
        if (!this.handle) return 0;
        return this.native.takeTriangleTests(this.handle);
    }

    sync() {
        this.native.sync(this.handle);
    }

    // The image of every frame submitted so far in canvas.pixels (waits for the GPU); returns the frameIndex it ends with.
    presentNow() {
        if (!this.accumulator || !this.canvas.pixels) return 0;
        return this.accumulator.render(null, this.canvas.pixels, true);
    }

    readAccum() {
        return this.native.readAccum(this.handle, this.resolution.width, this.resolution.height);
    }

    readIds() {
        return this.native.readPrimaryIds(this.handle, this.resolution.width, this.resolution.height);
    }

    readFrameF16() {
        return this.native.readFrameF16(this.handle, this.resolution.width, this.resolution.height);
    }

    // Multi-GPU row split (no counterpart in the reference): a renderer created with
    // { stripeIndex: i, stripeCount: n } renders the rows y with floor(y / stripeRows) % n == i.
    // ownedRows(): how many; readOwnedRows(): their accumulator values as three planes (R, G, B) of
    // ownedRows x width floats, in increasing y; importRows(block, i): on the renderer that assembles
    // the image, scatter the block read from the renderer with stripe index i into the accumulator.
    ownedRows() {
        return this.native.ownedRows(this.handle);
    }

    readOwnedRows() {
        return this.native.readOwnedRows(this.handle);
    }

    importRows(block, fromIndex) {
        this.native.writeRows(this.handle, block, fromIndex);
    }

    // Multi-GPU frame split (no counterpart in the reference): renderer r of n renders frames
    // first + r, first + r + n, ... of the whole image with renderFrameSums(first + r, count, n)
    // after clearAccum(); the planes then hold plain sums of the frames' colours.  The sums of all
    // renderers are added on the host (readAccum), written into one of them (writeAccum) and closed
    // with finishFrameSums(total number of frames).  Approximates the recurrence (include/rt_hip.h).
    clearAccum() {
        this.native.clearAccum(this.handle);
    }

    renderFrameSums(firstFrameIndex, nFrames, stride) {
        this.native.renderFrameSums(this.handle, firstFrameIndex, nFrames, stride || 1);
    }

    writeAccum(rgb) {
        this.native.writeAccum(this.handle, rgb);
    }

    finishFrameSums(nFramesTotal) {
        this.native.finishFrameSums(this.handle, nFramesTotal);
    }

    getCounters() {
        return this.native.readCounters(this.handle);
    }

    // GPU time of the most recent launch group and the number of frames it covered: render() calls made
    // while the GPU is busy are launched together, so the time of "the last frame" is ms / frames
    lastRenderMs() {
        return this.native.lastRenderMs(this.handle);
    }

    lastRenderFrames() {
        return this.native.lastRenderFrames(this.handle);
    }

    destroy() {
        this.stopRenderLoop();
        if (this.handle) this.native.destroy(this.handle);
        this.handle = null;
        this.accumulator = null;
    }
}

module.exports = { HipRenderer };
