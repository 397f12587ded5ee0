// Type declarations for accumulator.js: the surface of the reference's Accumulator (src/accumulator.ts:12-78).
export class Accumulator {
    /** `device` is the { native, handle } pair of a HipRenderer; the remaining arguments of the reference are ignored. */
    static create(device: { native: unknown; handle: unknown }, resolution: { width: number; height: number },
                  uniformsBuffer?: unknown, intermediateTexture?: unknown, sampler?: unknown): Promise<Accumulator>;
    /** canvasView: optional RGBA8 array of width*height*4 that receives the presented image. */
    render(commandEncoder: unknown, canvasView?: Uint8Array | Uint8ClampedArray | null): void;
}
