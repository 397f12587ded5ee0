'use strict';
/*
 * Accumulator: the interface of the reference's src/accumulator.ts (static
 * async create, render) over the native context.  The three r32float planes
 * and the running-mean pass of public/accumulator.wgsl live inside the native
 * context and run as part of each frame submission, so render() has nothing
 * left to encode; what it still does is the pass's other output, the 8-bit
 * canvas image (`return vec4(accumulatedColor, 1.0)`, accumulator.wgsl:67),
 * when the caller hands it a pixel array in place of a canvas view.
 */
class Accumulator {
    constructor(native, handle, resolution) {
        this.native = native;
        this.handle = handle;
        this.resolution = resolution;
    }

    // src/accumulator.ts:12-61.  `device` is the native module + context handle
    // pair; the buffer/texture/sampler arguments of the reference have no
    // counterpart (the native context owns them) and are accepted and ignored.
    static async create(device, resolution /*, uniformsBuffer, intermediateTexture, sampler */) {
        return new Accumulator(device.native, device.handle, resolution);
    }

    // src/accumulator.ts:63-78.  canvasView: optional Uint8Array / Uint8ClampedArray
    // of width*height*4 that receives the presented image (RGBA8, no gamma).
    render(commandEncoder, canvasView) {
        if (!canvasView) return;
        const w = this.resolution.width, h = this.resolution.height;
        const acc = this.native.readAccum(this.handle, w, h);
        for (let i = 0, n = w * h; i < n; i++) {
            // 8-bit unorm store of a value already clamped to [0, 1]
            canvasView[4 * i + 0] = Math.round(acc[3 * i + 0] * 255);
            canvasView[4 * i + 1] = Math.round(acc[3 * i + 1] * 255);
            canvasView[4 * i + 2] = Math.round(acc[3 * i + 2] * 255);
            canvasView[4 * i + 3] = 255;
        }
    }
}

module.exports = { Accumulator };
