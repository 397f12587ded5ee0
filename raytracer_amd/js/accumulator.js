'use strict';
/*
 * Accumulator: the interface of the reference's src/accumulator.ts (static
 * async create, render) over the native context.  The three r32float planes
 * and the running-mean pass of public/accumulator.wgsl live inside the native
 * context and run as part of each frame submission, so render() has nothing
 * left to encode; what it still does is ask for the pass's other output, the
 * 8-bit canvas image (`return vec4(accumulatedColor, 1.0)`,
 * accumulator.wgsl:67), when the caller hands it a pixel array in place of a
 * canvas view.
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

    // src/accumulator.ts:63-78.  canvasView: optional Uint8Array / Uint8ClampedArray of width*height*4 that
    // receives the presented image (RGBA8, no gamma), made on the GPU and copied to the host behind the frames
    // launched so far.  Like the reference's canvas pass this does not wait for the GPU: the array receives the
    // newest image that has arrived (none during the first frames); the return value is the frameIndex that image
    // ends with.  wait = true: the image of every frame submitted so far, waited for.
    render(commandEncoder, canvasView, wait) {
        if (!canvasView) return 0;
        return this.native.present(this.handle, canvasView, wait ? 1 : 0);
    }
}

module.exports = { Accumulator };
