// Type declarations for hip-renderer.js: the surface of the reference's WebGPURenderer
// (src/webgpu-renderer.ts:20-322) plus the read-back additions, for TypeScript hosts.
// The scene model is the reference's own (src/types.ts); a host inside that project imports its Scene type
// and may narrow `currentScene` to it.

export interface HipCanvas {
    width: number;
    height: number;
    /** Optional RGBA8 image of width*height*4 bytes that receives the presented frame. */
    pixels?: Uint8Array | Uint8ClampedArray;
}

export interface HipRendererOptions {
    device?: number;
    maxBounces?: number;
    stripeRows?: number;
    stripeIndex?: number;
    stripeCount?: number;
    /** Fill canvas.pixels on every k-th frame only (default 1). */
    presentEvery?: number;
    /** Apply a model's scale / rotation / position (the reference ignores them). Off by default. */
    applyModelTransforms?: boolean;
    /** Sample emissive spheres as well as emissive triangles at every bounce. Off by default. */
    sphereLights?: boolean;
    /** Make setDebug(true) draw every triangle's normal marker (common.wgsl:235-240). Off by default. */
    debugView?: boolean;
}

export class HipRenderer {
    private constructor(canvas: HipCanvas);
    /** Resolves to null on any failure, never rejects (src/webgpu-renderer.ts:48-103). */
    static Create(canvas: HipCanvas, scenePath: string, options?: HipRendererOptions): Promise<HipRenderer | null>;
    readonly currentScene: { camera: object; spheres: object[]; triangles: object[] } | null;
    frameIndex: number;
    setSamplesPerPixel(samples: number): void;
    setDebug(checked: boolean): void;
    resetAccumulation(): void;
    getFPS(): number;
    getEnabled(): boolean;
    setEnabled(value: boolean): void;
    startRenderLoop(): void;
    stopRenderLoop(): void;
    getTriangleTestsSinceLastCheck(): Promise<number>;
    render(): void;
    // additions for headless use
    renderFrames(n: number): void;
    sync(): void;
    readAccum(): Float32Array;
    readIds(): Int32Array;
    readFrameF16(): Uint16Array;
    getCounters(): number[];
    lastRenderMs(): number;
    presentNow(): number;
    lastRenderFrames(): number;
    clearAccum(): void;
    renderFrameSums(firstFrameIndex: number, nFrames: number, stride?: number): void;
    writeAccum(rgb: Float32Array): void;
    finishFrameSums(nFramesTotal: number): void;
    // multi-GPU row split: rows this renderer owns, their accumulator values (R, G, B planes), and assembly
    ownedRows(): number;
    readOwnedRows(): Float32Array;
    importRows(block: Float32Array, fromIndex: number): void;
    destroy(): void;
}
